"""The batch sizes `bench.py` times, held to the results the small-batch parity tests pin against the reference.

`tests/test_hip_fullsize.py` compares B = 1..2 evaluations with outputs captured from the reference itself.  The bench lines
are quoted at batch 256 (BASELINE config 2: ADM-G ImageNet-64, guided) and batch 64 (the 256x256 line): tile lists,
XCD orders, slab counts and 64-bit offsets there take values no small test sees (one adm256 activation is 2.1 GB).  The
engine's contract is that an image's result does NOT depend on how many images ride along (kernels and schedules are
chosen by shape, never by batch) -- so the first images of a bench-sized batch must equal, BITWISE, the small-batch
evaluation of the same inputs, which in turn is within the stated tolerance of the reference (asserted again here).
"""
import numpy as np
import pytest
import torch

from helpers import golden
from test_hip_fullsize import DEV, adm64, clf, guided_loop, load_filled, rel, u8_hist

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _pad_batch(t, n, seed, kind):
    """[k, ...] fixture tensor -> [n, ...]: the fixture rows first, seeded filler after them."""
    g = torch.Generator().manual_seed(seed)
    k = t.shape[0]
    if kind == "normal":
        fill = torch.randn((n - k,) + tuple(t.shape[1:]), generator=g)
    elif kind == "repeat":
        fill = t[torch.arange(n - k) % k]
    else:
        fill = torch.randint(0, kind, (n - k,), generator=g)
    return torch.cat([t, fill.to(t.dtype)], 0)


def test_adm64_batch_256_equals_the_batch_2_evaluation_bitwise():
    """BASELINE config 2 at its stated batch: UNet evaluation, guidance gradient and the guided 4-step loop at B = 256 --
    rows [0:2] bitwise equal to the B = 2 run that is checked against the reference's fp32 output."""
    B = 256
    g = golden("full_adm64")
    model, diffusion = adm64()
    x2, t2, y2 = (torch.from_numpy(g[k]) for k in ("x", "t", "y"))
    xb, tb, yb = _pad_batch(x2, B, 11, "normal"), _pad_batch(t2, B, 12, "repeat"), _pad_batch(y2, B, 13, 1000)
    out2 = model(x2.to(DEV), t2.to(DEV), y2.to(DEV))
    outb = model(xb.to(DEV), tb.to(DEV), yb.to(DEV))
    r = rel(out2, g["out"])
    print(f"ADM-G-64 UNet: B=2 vs reference fp32 {r:.3e}; B=256 rows [0:2] bitwise equal: {torch.equal(outb[:2], out2)}")
    assert r < 2e-2 and torch.isfinite(outb).all()
    assert torch.equal(outb[:2], out2)
    # a row in the middle and at the end of the big batch against its own B = 1 evaluation (other tiles / XCDs / slabs)
    for i in (129, B - 1):
        assert torch.equal(model(xb[i:i + 1].to(DEV), tb[i:i + 1].to(DEV), yb[i:i + 1].to(DEV)), outb[i:i + 1]), i

    gc = golden("full_clf64")
    c64 = clf(64, 4)
    xc, tc, yc = (torch.from_numpy(gc[k]) for k in ("x", "t", "y"))
    k = xc.shape[0]
    xcb, tcb, ycb = _pad_batch(xc, B, 21, "normal"), _pad_batch(tc, B, 22, "repeat"), _pad_batch(yc, B, 23, 1000)
    grad2, logits2 = c64.log_prob_grad(xc.to(DEV), tc.to(DEV), yc.to(DEV), 1.0, return_logits=True)
    gradb, logitsb = c64.log_prob_grad(xcb.to(DEV), tcb.to(DEV), ycb.to(DEV), 1.0, return_logits=True)
    rg = rel(grad2, gc["grad"])
    print(f"64x64 classifier: B={k} gradient vs reference autograd {rg:.3e}; B=256 rows [0:{k}] bitwise equal: "
          f"{torch.equal(gradb[:k], grad2)} (logits: {torch.equal(logitsb[:k], logits2)})")
    assert rg < 5e-2 and torch.isfinite(gradb).all()
    assert torch.equal(logitsb[:k], logits2)
    assert torch.equal(gradb[:k], grad2), float((gradb[:k] - grad2).abs().max() / grad2.abs().max())

    gl = golden("full_loop64")
    x_T, yl = torch.from_numpy(gl["x_T"]), torch.from_numpy(gl["y"])
    k = x_T.shape[0]
    x_Tb, ylb = _pad_batch(x_T, B, 31, "normal"), _pad_batch(yl, B, 33, 1000)
    s2, u2 = guided_loop(model, diffusion, c64, gl["cand"].tolist(), x_T, yl)
    sb, ub = guided_loop(model, diffusion, c64, gl["cand"].tolist(), x_Tb, ylb)
    rs = rel(s2, gl["ddim_g_sample"])
    h = u8_hist(ub[:k], gl["ddim_g_uint8"])
    print(f"guided 4-step DDIM loop (two streams): B={k} vs reference {rs:.3e}; B=256 rows bitwise equal: "
          f"{torch.equal(sb[:k], s2)}; uint8 of the B=256 rows vs the reference's within k levels {h}")
    assert rs < 2.5e-2 and torch.isfinite(sb).all() and ub.shape == (B, 64, 64, 3)
    assert torch.equal(sb[:k], s2) and torch.equal(ub[:k], u2)
    assert h[8] >= 0.99 and h[2] >= 0.90, h


@pytest.mark.parametrize("fixture,class_cond", [("full_lsun256", False), ("full_adm256cc", True)])
def test_lsun256_batch_64_equals_the_batch_1_evaluation_bitwise(fixture, class_cond):
    """The 256x256 bench line's batch (64; one bf16 activation of the first level is 2.1 GB > 2^31 bytes): rows of the big
    batch bitwise equal to their own B = 1 evaluations, the fixture row within tolerance of the reference -- for the
    unconditional LSUN network and for the class-conditional one BASELINE configs[4] names."""
    from bench import adm256_flags
    from autodiffusion_amd.script_util import create_model_and_diffusion
    B = 64
    g = golden(fixture)
    flags = adm256_flags()
    flags["class_cond"] = class_cond
    model, _ = create_model_and_diffusion(**flags)
    load_filled(model)
    x1, t1 = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    xb, tb = _pad_batch(x1, B, 41, "normal"), _pad_batch(t1, B, 42, "repeat")
    xb_d, tb_d = xb.to(DEV), tb.to(DEV)
    y1 = yb_d = None
    if class_cond:
        y1 = torch.from_numpy(g["y"]).to(DEV)
        yb_d = torch.cat([y1, torch.randint(0, 1000, (B - 1,), generator=torch.Generator().manual_seed(43)).to(DEV)])
    row = lambda v, i: None if v is None else v[i:i + 1]   # noqa: E731
    for tag, skip in (("out", []), ("out_skip", g["skip"].tolist())):
        outb = model(xb_d, tb_d, yb_d, skip_layer=skip)
        out1 = model(x1.to(DEV), t1.to(DEV), y1, skip_layer=skip)
        r = rel(out1[:, :, ::2, ::2], g[f"{tag}_sub"])
        k = x1.shape[0]
        print(f"{fixture} UNet ({tag}): B={k} vs reference {r:.3e}; B=64 rows [0:{k}] bitwise equal: {torch.equal(outb[:k], out1)}")
        assert r < 2e-2 and torch.isfinite(outb).all()
        assert torch.equal(outb[:k], out1)
        for i in (37, B - 1):
            assert torch.equal(model(xb_d[i:i + 1], tb_d[i:i + 1], row(yb_d, i), skip_layer=skip), outb[i:i + 1]), (tag, i)
        del outb
    torch.cuda.empty_cache()
