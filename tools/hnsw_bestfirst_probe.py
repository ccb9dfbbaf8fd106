#!/usr/bin/env python3
"""CPU probe for SURVEY 8f-2: would a best-first level-0 search (the standard algorithm, the shape of the
reference's unused src/scan/hnsw_scan.c) rescue recall on a graph built by the reference's hnswInsertNode?
Builds 20k x 64 with the oracle, searches with a plain best-first loop, reports recall and reachability.
Result (seed 0): recall@10 0.05 / 0.08 / 0.15 at ef 64 / 200 / 800, reference search 0.005, 5102 of 20000
nodes reachable from the entry point at level 0 - the links, not the search, are what is missing."""
import sys, time, heapq, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import ndbo
n, dim, m, efc = 20000, 64, 16, 200
rng = np.random.default_rng(0)
vecs = rng.standard_normal((n, dim)).astype(np.float32)
L = ndbo.lib()
g = ndbo.HnswGraph(dim, m=m, ef_construction=efc, cap_nodes=n+2)
t0=time.time()
for i in range(n):
    g.insert(vecs[i], i, L.ndbo_hnsw_level_from_uniform(float(rng.uniform(1e-9,1)), np.float32(0.36)))
print("build", time.time()-t0)
a = g.arrays()
nb0 = a["nbrs"][:,0,:]; cnt0 = a["ncount"][:,0]
V = a["vecs"]
def bestfirst(q, ef, k, entry):
    d0 = float(np.linalg.norm(V[entry]-q))
    cand=[(d0,entry)]; res=[(-d0,entry)]; vis={entry}
    while cand:
        d,c = heapq.heappop(cand)
        if len(res)>=ef and d > -res[0][0]: break
        for nb in nb0[c,:cnt0[c]]:
            nb=int(nb)
            if nb==0xFFFFFFFF or nb in vis or nb==0 or nb>n: continue
            vis.add(nb)
            dn=float(np.linalg.norm(V[nb]-q))
            if len(res)<ef or dn < -res[0][0]:
                heapq.heappush(cand,(dn,nb)); heapq.heappush(res,(-dn,nb))
                if len(res)>ef: heapq.heappop(res)
    return [b for _,b in sorted((-d,b) for d,b in res)][:k], len(vis)
Q = rng.standard_normal((100, dim)).astype(np.float32)
gt = np.argsort(((Q[:,None,:]-vecs[None,:,:])**2).sum(-1),axis=1)[:,:10]+1
for ef in (64, 200, 800):
    rec=0; ev=0
    for i,q in enumerate(Q):
        r,nv = bestfirst(q, ef, 10, a["entry_point"])
        rec += len(set(r)&set(gt[i].tolist()))/10; ev+=nv
    print("best-first on reference-built graph ef",ef,"recall",rec/100,"evals",ev/100)
rec=0
for i,q in enumerate(Q):
    eb,ed,ns = g.search(q,1,64,10)
    rec += len(set(eb.tolist())&set(gt[i].tolist()))/10
print("reference search recall", rec/100)
# reachability
seen={a["entry_point"]}; st=[a["entry_point"]]
while st:
    c=st.pop()
    for nb in nb0[c,:cnt0[c]]:
        nb=int(nb)
        if nb!=0xFFFFFFFF and nb not in seen and 0<nb<=n: seen.add(nb); st.append(nb)
print("reachable from entry at level 0:", len(seen), "of", n)
