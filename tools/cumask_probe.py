"""Which CUs does a CU-masked HIP stream get on this box?  (GPU; tools/r04 calls it.)

For a few masks: create the stream (adm_stream_create_cumask), launch the placement probe (adm_stream_probe: one block per slot,
each holding its CU for 200 us) and print how many distinct (XCC, SE, SH, CU) places and which XCCs the blocks landed on --
i.e. whether the mask is honoured at all, and whether mask bit i is CU i of XCD-major or of round-robin numbering."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
words = (ncu + 31) // 32


def stream_for(bits):
    m = (C.c_uint32 * words)()
    for i in bits:
        m[i // 32] |= 1 << (i % 32)
    out = C.c_void_p()
    _lib.check(lib.adm_stream_create_cumask(m, words, C.byref(out)), "create")
    return out.value


def probe(name, stream, nblocks=1024):
    out = torch.zeros(nblocks, dtype=torch.int32, device=dev)
    _lib.check(lib.adm_stream_probe(out.data_ptr(), nblocks, stream), "probe")
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype("uint32")
    xcc = v & 15
    hw = v >> 8
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    places = {(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(xcc, se, sh, cu)}
    per_xcc = {int(x): len({p for p in places if p[0] == x}) for x in sorted(set(xcc.tolist()))}
    print(f"{name:28s} distinct places {len(places):4d}  per XCC {per_xcc}", flush=True)


print(f"{torch.cuda.get_device_name(0)}: {ncu} CUs", flush=True)
probe("unmasked (null stream)", None)
cases = {
    "bits 0..127": range(128),
    "bits 128..255": range(128, 256),
    "bits 0..31": range(32),
    "bits i%8==0": [i for i in range(ncu) if i % 8 == 0],
    "bits i%8>=6 (64)": [i for i in range(ncu) if i % 8 >= 6],
    "bits >=192 (64)": range(192, 256),
    "bits >=160 (96)": range(160, 256),
}
for name, bits in cases.items():
    s = stream_for(list(bits))
    probe(name, s)
    _lib.check(lib.adm_stream_destroy(s), "destroy")
