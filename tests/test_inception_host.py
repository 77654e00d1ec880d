"""CPU: the Inception extractor's host surface (no GPU): parameter inventory against the CPU restatement, checkpoint key
layouts, pytorch_fid's constructor contract, and the loud failure without a GPU."""
import pytest
import torch

from autodiffusion_amd import inception
from autodiffusion_amd._lib import AdmError


def test_parameter_inventory_matches_the_restatement():
    from oracle import inception as oi
    assert dict(inception.param_shapes()) == oi.state_shapes()
    conv = sum(int(torch.tensor(s).prod()) for k, s in inception.param_shapes().items() if k.endswith("conv.weight"))
    assert conv == 21_751_136 and len(inception.CONVS) == 94      # Inception-v3 without the fc / AuxLogits heads


def test_checkpoint_key_layouts_and_strictness():
    m = inception.InceptionV3()
    assert not m.weights_loaded
    sd = {k: torch.full(s, 0.5) for k, s in inception.param_shapes().items()}
    full = dict(sd)
    full.update({"fc.weight": torch.zeros(1008, 2048), "fc.bias": torch.zeros(1008), "AuxLogits.fc.weight": torch.zeros(3, 3),
                 "Conv2d_1a_3x3.bn.num_batches_tracked": torch.tensor(0)})
    assert m.load_state_dict(full) == ([], []) and m.weights_loaded          # a full torchvision / pt_inception checkpoint
    assert float(m.state_dict()["Mixed_7c.branch_pool.bn.running_var"][0]) == 0.5
    with pytest.raises(RuntimeError, match="missing keys"):
        m.load_state_dict({k: v for k, v in sd.items() if not k.startswith("Mixed_6e")})
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(dict(sd, **{"Conv2d_1a_3x3.conv.weight": torch.zeros(32, 3, 5, 5)}))
    with pytest.raises(RuntimeError, match="unexpected keys"):
        m.load_state_dict(dict(sd, **{"Mixed_8a.conv.weight": torch.zeros(1)}))


def test_constructor_contract():
    assert inception.InceptionV3.BLOCK_INDEX_BY_DIM == {64: 0, 192: 1, 768: 2, 2048: 3}
    assert inception.InceptionV3([2]).last_needed_block == 2
    with pytest.raises(AssertionError):
        inception.InceptionV3([4])
    with pytest.raises(NotImplementedError):
        inception.InceptionV3(requires_grad=True)


def test_no_cpu_fallback():
    m = inception.InceptionV3()
    with pytest.raises(AdmError, match="GPU"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(AdmError, match="GPU"):
        m.features(torch.zeros(1, 64, 64, 3, dtype=torch.uint8))
