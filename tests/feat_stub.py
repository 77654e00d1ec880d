"""TEST stand-in for the Inception pool3 extractor (third-party, unavailable offline): a fixed random projection of the
uint8 image, as a `--features tests.feat_stub:factory` plug-in for scripts/search_ea.py."""
import torch


def factory(device):
    proj = (torch.randn(3 * 32 * 32, 24, generator=torch.Generator().manual_seed(5)) / 100.0).to(device)
    return (lambda u8: u8.reshape(u8.shape[0], -1).float() @ proj), 24
