#!/usr/bin/env python
"""bench.py -- mapped Gbp/s of the `-cx lr` seed-chain-align hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One STEP = one pass of the whole hot path (sketch -> seeds -> linear chaining -> graph chaining ->
WFA base alignment -> CIGAR/ds -> GAF text) over one batch of synthetic 10 kb ONT-like reads that is
already resident in HBM.  Workload = BASELINE.json configs[2]: 50 Mbp 3-haplotype bubble graph,
`-cx lr -c`.  Reads are sharded across ranks (each rank draws its own reads against the same graph:
weak scaling); the only inter-GPU traffic is the gather of GAF bytes to rank 0 over RCCL.

Rank 0 prints ONE JSON line (metric / roofline / cpu_baseline); everything else goes to stderr.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(ref_bin, graph, reads_fa, n_reads, out_dir, threads, cores):
    """the UNMODIFIED reference (oracle/_ref/minigraph) on a bounded sample of the same workload"""
    sample = os.path.join(out_dir, "cpu_sample.fa")
    with open(reads_fa) as fi, open(sample, "w") as fo:
        n = 0
        for line in fi:
            if line.startswith(">"):
                n += 1
                if n > n_reads:
                    break
            fo.write(line)
    n = min(n, n_reads)
    gaf, err = os.path.join(out_dir, "cpu.gaf"), os.path.join(out_dir, "cpu.log")
    with open(gaf, "wb") as fo, open(err, "w") as fe:
        subprocess.check_call([ref_bin, "-cx", "lr", "-t", str(threads), graph, sample], stdout=fo, stderr=fe)
    t_upd = t_map = None
    for line in open(err):
        m = re.match(r"\[M::(\w+)::([\d.]+)\*", line)
        if m:
            if m.group(1) == "mg_opt_update":
                t_upd = float(m.group(2))
            elif m.group(1) == "worker_pipeline":
                t_map = float(m.group(2))
    bases = sum(len(l.strip()) for l in open(sample) if not l.startswith(">"))
    dt = (t_map - t_upd) if (t_upd is not None and t_map is not None) else float("nan")
    return dict(value=bases / dt / 1e9, unit="Gbp/s", cores=cores, kind="reference",
                sample="%d reads (%d bp) of the same workload, minigraph -cx lr -t %d on %d usable cores (affinity capped by the cgroup CPU quota), map phase only (worker_pipeline - mg_opt_update)" % (n, bases, threads, cores)), gaf, n


def usable_cores():
    """cores this process may actually burn: the affinity mask capped by the cgroup CPU quota (cpu.max)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def cgroup_throttled():
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
    except Exception:
        return 0, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU per step (BASELINE configs[2]: 100k reads)")
    ap.add_argument("--genome", type=int, default=50000000)
    ap.add_argument("--hap", type=int, default=3)
    ap.add_argument("--cpu-reads", type=int, default=20000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--gather", action="store_true", help="N>1: also gather every rank's GAF bytes to rank 0 over RCCL inside the timed region "
                    "(off by default: the shards are independent, each rank keeps / writes its own GAF)")
    ap.add_argument("--threads", type=int, default=0, help="host threads per rank (0: min(64, usable cores / ranks))")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log("[bench] WORLD_SIZE=%d overrides --gpus %d" % (world, args.gpus))
    n_gpus = world
    os.environ.setdefault("MGA_DEVICE", str(local_rank))

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import minigraph_amd as mga
    from minigraph_amd.dist import gather_bytes
    L = mga.load()
    if L.mga_device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    ncpu = os.cpu_count() or 1
    quota = usable_cores()
    threads = args.threads or max(2, min(64, quota // world))  # [measured] more threads than usable cores only adds contention + CFS throttling

    # ---- synthetic workload (untimed) ----
    d = tempfile.mkdtemp(prefix="mga_bench_r%d_" % rank)
    pre = os.path.join(d, "w")
    t0 = time.time()
    subprocess.check_call([mga.MGSIM, "-p", pre, "-G", str(args.genome), "-H", str(args.hap), "-n", str(args.reads),
                           "-s", "11", "-S", str(1000 + rank)], stderr=subprocess.DEVNULL)
    t_gen = time.time() - t0
    graph_path, reads_path = pre + ".gfa", pre + ".reads.fa"
    t0 = time.time()
    G = mga.Graph(graph_path, preset="lr", cigar=True, n_threads=threads)
    t_index = time.time() - t0
    R = mga.Reads(reads_path)
    log("[bench] rank %d: gen %.1fs, load+index %.1fs, %d reads / %d bp resident, %d host threads" % (rank, t_gen, t_index, R.n, R.bases, threads))

    def step():
        gaf = mga.map_reads(G, R, n_threads=threads, copy=False)   # GAF text stays in the C library's buffer
        if dist is not None and args.gather:  # optional: one output stream on rank 0; zero-copy view -> GPU -> RCCL gather
            gather_bytes(gaf.view(), dst=0, device="cuda", as_tensors=True)
        return gaf

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gaf = None
    for _ in range(args.warmup):
        gaf = step()
    mga.get_stats(G, reset=True)
    mga.prof_enable(True)
    mga.prof_get(reset=True)
    sync()
    cpu0, thr0 = time.process_time(), cgroup_throttled()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gaf = step()
    torch.cuda.synchronize()
    sync()
    dt = time.perf_counter() - t0
    cpu1, thr1 = time.process_time(), cgroup_throttled()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tb = torch.tensor([R.bases], dtype=torch.int64, device="cuda")
        dist.all_reduce(tb)
        total_bases = int(tb.item())
    else:
        total_bases = R.bases
    st, prof = mga.get_stats(G), mga.prof_get()

    if rank == 0:
        value = total_bases * args.steps / dt / 1e9
        # dominant kernel by HIP-event time on its launch stream
        fam = {"k_wfa": sum(prof[k][0] for k in prof if k.startswith("k_wfa")), "k_sketch": prof["k_sketch"][0],
               "k_seed": prof["k_seed_count"][0] + prof["k_seed_fill"][0], "k_lchain": prof["k_lchain"][0]}
        launches = {"k_wfa": sum(prof[k][1] for k in prof if k.startswith("k_wfa")), "k_sketch": prof["k_sketch"][1],
                    "k_seed": prof["k_seed_count"][1] + prof["k_seed_fill"][1], "k_lchain": prof["k_lchain"][1]}
        alg = {  # algorithmic bytes of each kernel family over the timed steps
            "k_wfa": st["wfa_t_bases"] + st["wfa_q_bases"],
            "k_sketch": 2 * st["n_bases"] + 16 * st["n_mz"],           # two passes over the bases (count, write) + minimizers out
            "k_seed": 16 * st["n_probe"] + 8 * st["n_hit"] + 16 * st["n_hit"],
            "k_lchain": 32 * st["n_hit"],
        }
        dom = max(fam, key=lambda k: fam[k])
        # HBM traffic of the dominant family per launch, from the committed PMC passes of this same command (profiles/*_pmc.json;
        # bench.py cannot run rocprofv3 around itself).  KB of FETCH_SIZE + WRITE_SIZE, uncorrected (see the file's "source").
        traffic, traffic_src = None, None
        try:
            import glob
            pf = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))[-1]
            pk = json.load(open(pf))["kernels"]
            sel = [v for k, v in pk.items() if (k == "k_wfa" or k.startswith("k_wfa_r<") if dom == "k_wfa" else k.startswith(dom))]  # the tier kernels, not the scheduler's helpers
            nl = sum(v.get("launches_fetch", 0) for v in sel)
            if nl:
                traffic = sum(v.get("fetch_kb", 0) + v.get("write_kb", 0) for v in sel) * 1024.0 / nl
                traffic_src = os.path.relpath(pf, ROOT)
        except Exception:
            pass
        ach = alg[dom] / (fam[dom] * 1e-3) / 1e9 if fam[dom] > 0 else 0.0
        roof = dict(bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                    launches=launches[dom], avg_launch_ms=fam[dom] / max(1, launches[dom]),
                    alg_bytes_per_launch=alg[dom] / max(1, launches[dom]))
        res = dict(metric="mapped Gbp/sec (whole node), -cx lr 10kb reads, graph base alignment (-c)", value=value, unit="Gbp/s",
                   n_gpus=n_gpus, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="u8/int32", data="synthetic",
                   config=dict(workload="configs[2]: %d x 10kb synthetic ONT reads per GPU vs %d bp %d-haplotype bubble graph, -cx lr -c"
                               % (R.n, args.genome, args.hap), reads_per_gpu=R.n, read_len=10000, err=0.1, sharding="reads/%dgpu, no data-path collective%s" % (n_gpus, " + RCCL gather of GAF bytes" if (args.gather and n_gpus > 1) else ""),
                               host_threads_per_rank=threads),
                   roofline=roof,
                   kernels_ms={k: round(v[0], 3) for k, v in prof.items()},
                   stage_s={k: round(v, 4) for k, v in st.items() if k.startswith("t_")},
                   per_read=dict(n_mz=st["n_mz"] / max(1, st["n_reads"]), n_hit=st["n_hit"] / max(1, st["n_reads"]),
                                 n_wfa=st["n_wfa"] / max(1, st["n_reads"]), wfa_cells=st["wfa_cells"] / max(1, st["n_reads"]),
                                 gaf_bytes=st["gaf_bytes"] / max(1, st["n_reads"])),
                   index_s=round(t_index, 2),
                   host=dict(logical_cpus=ncpu, usable_cores=quota, cpu_s_per_step=round((cpu1 - cpu0) / args.steps, 3),
                             cfs_throttled_periods=thr1[0] - thr0[0], cfs_throttled_s=round((thr1[1] - thr0[1]) * 1e-6, 3)))
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "minigraph")
        if not args.no_cpu and os.path.exists(ref_bin) and n_gpus == 1:
            try:
                cb, cpu_gaf, n_cpu = cpu_baseline(ref_bin, graph_path, reads_path, args.cpu_reads, d, min(ncpu, 2 * quota), quota)
                res["cpu_baseline"] = cb
                want = open(cpu_gaf, "rb").read()
                gaf = gaf.bytes()
                res["parity"] = "GAF byte-identical to the reference on the %d-read sample" % n_cpu if gaf[:len(want)] == want and (len(gaf) == len(want) or gaf[len(want) - 1:len(want)] == b"\n") else "MISMATCH vs reference GAF"
            except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
                res["cpu_baseline"] = dict(value=None, unit="Gbp/s", cores=quota, kind="reference", sample="failed: %r" % (e,))
        else:
            res["cpu_baseline"] = dict(value=None, unit="Gbp/s", cores=quota, kind="reference",
                                       sample="not run (N>1, --no-cpu, or oracle/_ref/minigraph absent)")
        print(json.dumps(res), flush=True)
    R.close()
    G.close()
    shutil.rmtree(d, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
