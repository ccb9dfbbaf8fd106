#!/usr/bin/env python3
"""C5-shaped check: fp16 rows, inner product, B = 256 — scan mode 5 (fp16 matrix-core screen) against mode 2
(exact grouped scan) on the same mirror; prints mismatching queries and the library statistics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import make_data, pack_tids, unpack_tids

def main():
    from neurondb_amd import IvfIndex, _lib
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dim, lists, nq, K, P = 1536, int(sys.argv[2]) if len(sys.argv) > 2 else 1024, 256, 10, 32
    strategy = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0); _lib.use_torch_stream()
    lib, check = _lib.lib(), _lib.check
    base = make_data(n, dim, "clustered", lists, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    q = make_data(nq, dim, "clustered", lists, 0.1, 0x5EED0002, 0x5EEDC0DE, dev)
    ix = IvfIndex(dim, lists)
    ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
    twin = ix.to_f16(False); ix.close(); ix = twin
    res = {}
    for mode in (2, 5, 3):
        check(lib.ndbhip_set_scan_mode(mode)); check(lib.ndbhip_stats_reset())
        ot = torch.zeros((nq, K), dtype=torch.int64, device=dev)
        od = torch.zeros((nq, K), dtype=torch.float32, device=dev)
        oc = torch.zeros(nq, dtype=torch.int32, device=dev)
        ix.search_device(q, ot, od, oc, strategy, P, K, 0)
        check(lib.ndbhip_synchronize())
        res[mode] = (unpack_tids(ot).cpu().numpy(), od.cpu().numpy())
        print("mode", mode, {k: v for k, v in _lib.stats().items() if k in ("rows_rescored", "rows_emitted", "screen16_batches", "screen16_fallbacks")})
    for mode in (5, 3):
        bad = [i for i in range(nq) if not (np.array_equal(res[2][0][i], res[mode][0][i]) and
                                            np.array_equal(res[2][1][i].view(np.uint32), res[mode][1][i].view(np.uint32)))]
        print("mode", mode, "mismatching queries:", len(bad), bad[:10])
        for i in bad[:2]:
            print("  exact:", res[2][0][i], res[2][1][i]); print("  got  :", res[mode][0][i], res[mode][1][i])

if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "replay"):
    main()


def replay():
    """oracle replay of a few queries over the probed lists only (the logic of tests/test_gpu_fullsize.py)"""
    import ctypes as C
    from neurondb_amd import IvfIndex, _lib
    from oracle import ndbo
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    dim, lists, nq, K, P, strategy = 1536, (4096 if n > 2_000_000 else 1024), 256, 10, 32, 3
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0); _lib.use_torch_stream()
    lib, check = _lib.lib(), _lib.check
    base = make_data(n, dim, "clustered", lists, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    q = make_data(nq, dim, "clustered", lists, 0.1, 0x5EED0002, 0x5EEDC0DE, dev)
    ix = IvfIndex(dim, lists)
    ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
    f16 = not (len(sys.argv) > 3 and sys.argv[3] == "f32")
    strategy = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    if len(sys.argv) > 5:
        check(lib.ndbhip_set_scan_mode(int(sys.argv[5])))
    if f16:
        twin = ix.to_f16(False); ix.close(); ix = twin
        for s0 in range(0, n, 1 << 20):
            h = base[s0:s0 + (1 << 20)].to(torch.float16); f = h.to(torch.float32)
            sub = (h.abs() < 2.0 ** -14) & (h != 0)
            base[s0:s0 + (1 << 20)] = torch.where(sub, f * 2.0 ** -10, f)
    ot = torch.zeros((nq, K), dtype=torch.int64, device=dev)
    od = torch.zeros((nq, K), dtype=torch.float32, device=dev)
    oc = torch.zeros(nq, dtype=torch.int32, device=dev)
    ix.search_device(q, ot, od, oc, strategy, P, K, 0)
    check(lib.ndbhip_synchronize())
    rows, dist = unpack_tids(ot).cpu().numpy(), od.cpu().numpy()
    qh = q.cpu().numpy()
    cent_h, ll, _, _ = ix.export(rows=False)
    t6 = np.zeros((n, 6), np.uint8)
    check(lib.ndbhip_ivf_export(ix._h, None, None, None, t6.ctypes.data_as(C.c_void_p)))
    tid_all = t6.view(ndbo.TID_DTYPE).reshape(n)
    order = (((tid_all["bi_hi"].astype(np.int64) << 16) | tid_all["bi_lo"]) * 64 + tid_all["posid"] - 1)
    print("order is a permutation:", len(np.unique(order)) == n, order[:5], ll[:5], ll.sum())
    off = np.zeros(len(ll) + 1, np.int64); off[1:] = np.cumsum(ll)
    for i in (0, 100):
        pr = sorted(int(x) for x in ix.select_clusters(qh[i:i + 1], P)[0] if x >= 0)
        keep = np.zeros(len(ll), bool); keep[pr] = True
        ll2 = np.where(keep, ll, 0); off2 = np.zeros(len(ll) + 1, np.int64); off2[1:] = np.cumsum(ll2)
        sel = np.concatenate([np.arange(off[L], off[L + 1]) for L in pr])
        rows_img = base[torch.from_numpy(order[sel]).to(dev)].cpu().numpy()
        img = ndbo.IvfImage(cent_h, off2, rows_img, np.ascontiguousarray(tid_all[sel]))
        et, ed, _ = img.search(qh[i], strategy, P, K, 0)
        erow = ((et["bi_hi"].astype(np.int64) << 16) | et["bi_lo"]) * 64 + et["posid"] - 1
        print(i, "oracle", erow, ed); print(i, "gpu   ", rows[i], dist[i])
        allrows = torch.from_numpy(order[sel]).to(dev)
        ip = -(base[allrows].double() @ q[i].double()) if strategy == 3 else ((base[allrows].double() - q[i].double()) ** 2).sum(1).sqrt()
        top = torch.topk(ip, K, largest=False)
        print(i, "torch ", allrows[top.indices].cpu().numpy(), top.values.cpu().numpy(), "candidates", len(sel), "lists", [(L, int(ll[L])) for L in pr][:6])
        print(i, "oracle probes", [int(x) for x in img.select_clusters(qh[i], P)][:8], "gpu probes", [int(x) for x in ix.select_clusters(qh[i:i+1], P)[0]][:8])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "replay":
    replay()
