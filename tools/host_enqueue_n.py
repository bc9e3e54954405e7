#!/usr/bin/env python
"""Host cost of enqueueing one training step with N rank processes running CONCURRENTLY on this host (VERDICT r3 #7d): N independent
processes (no process group: a gloo all-reduce would block the host and hide what is being measured), each pinned as bench.py pins its
ranks, all sharing the one visible GPU.  The device is N x slower per process, but the number reported is HOST time: wall time of
TrainStep.step() returning, queue drained before each step.  usage: python tools/host_enqueue_n.py N [precision]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == '--worker':
    rank, n, prec = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import bench
    from warpedganspace_amd.hostpin import plan
    cpus = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, plan(cpus, cpus, n, rank))          # a disjoint slice per process (one NUMA-node slice per rank on an 8-GPU box)
    dev = torch.device('cuda:0')
    eng = bench.build(dev, 'stylegan2', 128, 32, 32, rank=rank, precision=prec)
    for _ in range(4):
        eng.step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step()
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    ts.sort()
    print(json.dumps({"rank": rank, "min_ms": round(ts[0], 3), "median_ms": round(ts[len(ts) // 2], 3), "max_ms": round(ts[-1], 3), "cpus": len(os.sched_getaffinity(0))}), flush=True)
    sys.exit(0)

n = int(sys.argv[1]); prec = sys.argv[2] if len(sys.argv) > 2 else 'auto'
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker', str(r), str(n), prec], stdout=subprocess.PIPE, text=True) for r in range(n)]
rows = []
for p in procs:
    out = p.communicate()[0]
    rows += [json.loads(l) for l in out.splitlines() if l.startswith('{')]
rows.sort(key=lambda r: r['rank'])
print(json.dumps({"processes": n, "precision": prec, "host_enqueue_ms_per_step": {"min_over_ranks_of_median": min(r['median_ms'] for r in rows),
                  "max_over_ranks_of_median": max(r['median_ms'] for r in rows), "worst_single_step": max(r['max_ms'] for r in rows)}, "ranks": rows}))
