"""Which compute units does a CU-masked stream (ops.masked_stream) dispatch to?  n one-wave workgroups that spin 20 us each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from semireward_amd import ops

dev = "cuda:0"
for ncu in [int(a) for a in sys.argv[1:]] or [0, 192, 128, 64]:
    st = ops.masked_stream(ncu, dev) if ncu else torch.cuda.Stream(dev)
    n = 4096
    out = torch.zeros(n, 2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        ops._call("srhip_cu_probe", out.data_ptr(), n, 2000, st.cuda_stream)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    xcc, hw = o[:, 0], o[:, 1]
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    print("mask = first %d CU bits: %d distinct (xcd, se, sh, cu) used" % (ncu, len(set(zip(xcc, se, sh, cu)))))
    for x in sorted(set(xcc)):
        m = xcc == x
        units = sorted(set(zip(se[m], sh[m], cu[m])))
        print("  XCD %d: %4d workgroups on %2d CUs; per SE: %s" % (x, m.sum(), len(units), {int(s): sum(1 for u in units if u[0] == s) for s in sorted(set(se[m]))}))
