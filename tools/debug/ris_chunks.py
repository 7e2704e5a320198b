#!/usr/bin/env python3
"""Device-resident fused Ristretto round trip (one launch over the table ring) against the host-buffer
path of the same library (2^18-element chunks, one stream) and the oracle on a sample."""
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dusk_zerocaf_amd as z
from oracle import zc_ref
from tests.vectors import rand_scalars_np

zc_ref.build()
eng = z.Engine([0])
eng.set_stream(torch.cuda.current_stream().cuda_stream)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
res = {}
for lg in (20, 22):
    n = 1 << lg
    P = eng.ed_mul_base(dev(rand_scalars_np(n, 11, 249)))
    K = rand_scalars_np(n, 12, 252)
    Kd = dev(K)
    enc = eng.ris_compress(P)
    torch.cuda.synchronize()
    enc[::97, 31] |= 0x80
    torch.cuda.synchronize()
    out = torch.empty_like(enc)
    for rep in range(3):
        o, ok = eng.ris_roundtrip_mul(enc, Kd, out=out)
        torch.cuda.synchronize()
        h_enc = enc.cpu().numpy()
        ho, hok = eng.ris_roundtrip_mul(h_enc, K)
        d_o, d_ok = o.cpu().numpy(), ok.cpu().numpy()
        bad = np.nonzero((d_o != ho).any(axis=1) | (d_ok != hok))[0]
        res["2p%d_rep%d" % (lg, rep)] = {"mismatch": int(len(bad)), "first": bad[:8].tolist(), "last": bad[-8:].tolist(),
                                         "chunks": sorted(set((bad // 786432).tolist()))}
    m = 4096
    wo, wok = zc_ref.ris_roundtrip_mul(h_enc[:m], K[:m])
    res["2p%d_host_vs_oracle" % lg] = bool(np.array_equal(wo, ho[:m]) and np.array_equal(wok, hok[:m]))
    res["2p%d_dev_vs_oracle" % lg] = bool(np.array_equal(wo, d_o[:m]) and np.array_equal(wok, d_ok[:m]))
print(json.dumps(res))
