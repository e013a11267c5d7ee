#!/usr/bin/env python
"""bench.py -- mapped Gbp/s of the `-cx lr` seed-chain-align hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload = the configuration BASELINE.json's metric is quoted on (configs[3] shape): a 3 Gbp 5-haplotype bubble graph in 24
chromosomes and 125 000 x 10 kb synthetic ONT reads PER GPU (1 M reads at 8 GPUs), `-cx lr -c`.

One STEP = one pass of the whole mapping phase over that read set: FASTA file -> parse -> H2D -> sketch -> seeds -> linear chaining ->
graph chaining -> WFA base alignment -> CIGAR/ds -> GAF text in ONE buffer in memory -- the interval between the reference's
`mg_opt_update` and its last `worker_pipeline` log line (gmap.c:186-211), which is also what the
`cpu_baseline` leg reports for the unmodified reference.  Graph load and index build are outside it on both sides (`index_s`).
N > 1: ONE read file (125 000 x N reads, written by rank 0) is cut into N contiguous byte ranges by the library's reader, every rank
maps its range against its own replica of the index, and the GAF bytes are gathered to rank 0 over RCCL and re-assembled in input
order inside the timed region.  Extra keys: `resident` (the same reads already in HBM, no parse / upload), `isolated`
(one pass with ONE chunk in flight, so that per-kernel HIP-event times do not overlap: the per-kernel roofline comes from there).

`python bench.py --gpus N` without a launcher around it starts its own N ranks (torch.distributed.run); a WORLD_SIZE that disagrees with --gpus is refused.
Graph chaining + the gap list run on the host threads or on the device (the library picks by the host threads a rank has): `value` is the library's choice,
`device_placement` / `host_placement` is the other one, timed the same way right after it, both compared byte for byte with the reference on every read of the sample.

N = 1 also runs, after the headline, the other BASELINE configurations at their sizes, each compared byte for byte with the reference on the same box: `file_out` (the headline's step with the
GAF written to a file, the interval of the CPU leg), `linear` / `bubble50` (configs[1] / configs[2]: 100 000 reads vs the 50 Mbp linear FASTA / 3-haplotype graph) and `asm`
(configs[4] at one GPU's share: -cx asm, one ~98 Mbp contig per chromosome vs the 3 Gbp graph, then --call through the reference's own caller).

Rank 0 prints ONE JSON line (metric / roofline / cpu_baseline); everything else goes to stderr.
"""
import argparse
import ctypes
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
METRIC = "mapped Gbp/sec (whole node), -cx lr 10kb reads vs 3Gbp graph, 1/2/4/8 GPU"  # BASELINE.json, verbatim


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
    dt, t_idx = ref_map_phase(err)
    bases = sum(len(l.strip()) for l in open(sample) if not l.startswith(">"))
    return dict(value=bases / dt / 1e9, unit="Gbp/s", cores=cores, kind="reference", index_s=round(t_idx, 1),
                sample="%d reads (%d bp) of the same workload, minigraph -cx lr -t %d on %d usable cores (affinity capped by the cgroup CPU quota), map phase only (worker_pipeline - mg_opt_update: FASTA parse + mapping + GAF write, as in `value`)" % (n, bases, threads, cores)), gaf, n


def kernel_src_sha1():
    """hash of every device source of this tree -- the same function as minigraph_amd/tools/prof_summary.py's, which stamps it into the counter / PMC summaries"""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "minigraph_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "minigraph_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()


def ref_map_phase(log_path):
    """seconds of the reference's mapping phase (worker_pipeline - mg_opt_update: SURVEY 8d) and of its graph load + index, from its own log lines"""
    ts = {}
    for line in open(log_path):
        m = re.match(r"\[M::(\w+)::([\d.]+)\*", line)
        if m:
            ts[m.group(1)] = float(m.group(2))
    t_upd, t_map = ts.get("mg_opt_update"), ts.get("worker_pipeline")
    return ((t_map - t_upd) if (t_upd is not None and t_map is not None) else float("nan")), ts.get("mg_opt_update", float("nan"))


def asm_block(mga, d, wl, graph_path, threads, ref_bin, dropin_bin, n_contig=0, small_genome=0):
    """BASELINE configs[4] at ONE GPU's share: `-cx asm` on chromosome-scale contigs (one per chromosome, 0.1 %% divergence from a haplotype walk) against the SAME 3 Gbp graph the
    headline maps reads against; the whole job file -> file (GFA parse, index, mapping, GAF) here and in the unmodified reference on the same cores: the two GAF files must be the same
    bytes.  Then `--call` (ggen.c:128-139 -> mg_call_asm, asm-call.c:21): the reference's own front end linked against THIS library (oracle/_ref/minigraph_dropin: its ggen.c / asm-call.c
    consuming our mg_gchains_t) against the all-reference binary: the two BED files must be the same bytes.
    --asm-genome N (N > 0): a smaller self-made graph of N backbone bp in 10 chromosomes with 10 contigs (quick runs)."""
    pre = os.path.join(d, "asm")
    if small_genome > 0:
        genome, n_chr, hap, n = small_genome, 10, 3, 10
        contig = genome // 10
        subprocess.run([mga.MGSIM, "-p", pre, "-G", str(genome), "-c", "10", "-H", "3", "-n", "10", "-l", str(contig), "-e", "0.001", "-s", "5"], stderr=subprocess.DEVNULL, check=True)
        g = pre + ".gfa"
    else:   # the headline's graph (same -G -c -H -s => the same bytes; -R: only the contigs are written), contigs from their own stream
        genome, n_chr, hap = wl["genome"], wl["chr"], wl["hap"]
        n = n_contig or n_chr
        contig = genome // n_chr - 1000
        subprocess.run([mga.MGSIM, "-R", "-p", pre, "-G", str(genome), "-c", str(n_chr), "-H", str(hap), "-s", "11", "-S", "5", "-n", str(n), "-l", str(contig), "-e", "0.001"], stderr=subprocess.DEVNULL, check=True)
        g = graph_path
    r, got, ref, got_bed, ref_bed, log_ = pre + ".reads.fa", pre + ".got.gaf", pre + ".ref.gaf", pre + ".got.bed", pre + ".ref.bed", pre + ".ref.log"
    q_bp = sum(len(l) - 1 for l in open(r) if not l.startswith(">"))
    try:
        mga.load().mga_rq_dev_stats((ctypes.c_int64 * 8)(), 1)
    except Exception:
        pass
    t0 = time.time()
    mga.map_files(g, [r], got, preset="asm", cigar=True, n_threads=threads, verbose=0)
    t_ours = time.time() - t0
    out = dict(workload="configs[4] shape at one GPU's share: -cx asm, %d contigs x %.1f Mbp (%.2f Gbp of query, 0.1%% divergence from the haplotype walks) vs the %.2f Gbp-backbone %d-haplotype bubble graph in %d chromosomes%s"
                        % (n, contig / 1e6, q_bp / 1e9, genome / 1e9, hap, n_chr, "" if small_genome > 0 else " (the headline's graph)"),
               interval="file -> file: GFA parse + index + mapping + GAF, on both sides", seconds=round(t_ours, 2), query_Mbp_per_s=round(q_bp / 1e6 / t_ours, 1),
               gaf_bytes=os.path.getsize(got), host_threads=threads,
               note="the forward passes of the primary chainer under -x asm (mg_lchain_rmq, lchain.c:252-357) run on the device, a wavefront per (segment, strand) run (k_rmq.hip); "
                    "its anchor sort (klib's exact permutation) and backtrack on the host threads; sketch, seeds, WFA and text on the device")
    try:   # which runs of the RMQ chainer the device took, and which it handed back to the host's exact tree (tied priorities / inner window beyond the kernel's sort / too long)
        st_rq = (ctypes.c_int64 * 8)()
        mga.load().mga_rq_dev_stats(st_rq, 1)
        out["rmq_runs"] = dict(device=int(st_rq[0]), host_tie=int(st_rq[1]), host_inner_window=int(st_rq[2]), host_long=int(st_rq[3]), host_device_failed=int(st_rq[4]))
    except Exception:
        pass
    if ref_bin and os.path.exists(ref_bin):
        t0 = time.time()
        with open(ref, "wb") as fo:
            subprocess.run([ref_bin, "-c", "-x", "asm", "-t", str(threads), g, r], stdout=fo, stderr=subprocess.DEVNULL, check=True)
        t_ref = time.time() - t0
        out["reference_seconds"] = round(t_ref, 2)
        out["vs_reference"] = round(t_ref / t_ours, 2)
        out["parity"] = "GAF byte-identical to the reference" if subprocess.call(["cmp", "-s", got, ref]) == 0 else "MISMATCH vs reference GAF"
        if dropin_bin and os.path.exists(dropin_bin):   # --call: BED of the reference's own caller on OUR chains vs on its own
            t0 = time.time()
            with open(got_bed, "wb") as fo:
                rc = subprocess.run([dropin_bin, "-cx", "asm", "--call", "-t", str(threads), g, r], stdout=fo, stderr=subprocess.DEVNULL).returncode
            t_call = time.time() - t0
            t0 = time.time()
            with open(ref_bed, "wb") as fo, open(log_, "w") as fe:
                subprocess.run([ref_bin, "-cx", "asm", "--call", "-t", str(threads), g, r], stdout=fo, stderr=fe, check=True)
            t_call_ref = time.time() - t0
            out["call"] = dict(seconds=round(t_call, 2), reference_seconds=round(t_call_ref, 2), vs_reference=round(t_call_ref / max(t_call, 1e-9), 2), bed_bytes=os.path.getsize(ref_bed),
                               command="minigraph -cx asm --call: main.c / ggen.c / asm-call.c / gfa-bbl.c of the reference linked against libminigraph_amd.so (oracle/_ref/minigraph_dropin, mg_map_batch patch of INTEGRATION.md 1b) vs the unmodified binary")
            out["call_parity"] = ("BED byte-identical to the reference (%d bytes)" % os.path.getsize(ref_bed)) if (rc == 0 and os.path.getsize(ref_bed) > 0 and subprocess.call(["cmp", "-s", got_bed, ref_bed]) == 0) \
                else "MISMATCH vs reference BED (dropin exit code %d)" % rc
        else:
            out["call_parity"] = "not run: oracle/_ref/minigraph_dropin absent"
    for f in (r, got, ref, got_bed, ref_bed, log_, pre + ".lin.fa") + ((pre + ".gfa",) if small_genome > 0 else ()):
        try:
            os.remove(f)
        except OSError:
            pass
    return out


def small_configs_block(mga, d, threads, ref_bin, n_reads, genome):
    """BASELINE configs[1] and configs[2] at their sizes on one GPU: the SAME n_reads x 10 kb reads (a) vs the 50 Mbp backbone as a linear FASTA -- ONE segment of 50 Mbp, `-x lr`
    without base alignment (minimap2-like) -- and (b) vs the 50 Mbp 3-haplotype bubble graph, `-cx lr`.  Graph load + index outside the interval on both sides (our index_s / the
    reference's up to mg_opt_update), mapping phase file -> GAF FILE on both sides; the GAF files must be the same bytes."""
    pre = os.path.join(d, "c12")
    subprocess.run([mga.MGSIM, "-p", pre, "-G", str(genome), "-c", "1", "-H", "3", "-n", str(n_reads), "-s", "11"], stderr=subprocess.DEVNULL, check=True)
    rd = pre + ".reads.fa"
    bases = sum(len(l) - 1 for l in open(rd) if not l.startswith(">"))
    res = {}
    for key, graph, cigar, what in (("linear", pre + ".lin.fa", False, "configs[1]: %d x 10kb reads vs a %.0f Mbp linear FASTA (one segment), -x lr" % (n_reads, genome / 1e6)),
                                    ("bubble50", pre + ".gfa", True, "configs[2]: the same reads vs the %.0f Mbp-backbone 3-haplotype bubble graph, -cx lr" % (genome / 1e6))):
        got, ref, log_ = pre + "." + key + ".got.gaf", pre + "." + key + ".ref.gaf", pre + "." + key + ".ref.log"
        t0 = time.time()
        G = mga.Graph(graph, preset="lr", cigar=cigar, n_threads=threads)
        t_idx = time.time() - t0
        mga.map_files_idx(G, [rd], n_threads=threads, out_path=got)   # (warm-up: pipeline contexts, buffers)
        t_map = mga.map_files_idx(G, [rd], n_threads=threads, out_path=got)
        G.close()
        o = dict(workload=what, value=bases / t_map / 1e9, unit="Gbp/s", map_seconds=round(t_map, 3), index_s=round(t_idx, 2), gaf_bytes=os.path.getsize(got),
                 interval="FASTA file -> GAF file, mapping phase (graph load + index outside, on both sides)")
        if ref_bin and os.path.exists(ref_bin):
            with open(ref, "wb") as fo, open(log_, "w") as fe:
                subprocess.run([ref_bin] + (["-c"] if cigar else []) + ["-x", "lr", "-t", str(threads), graph, rd], stdout=fo, stderr=fe, check=True)
            t_ref, t_ref_idx = ref_map_phase(log_)
            o["reference"] = dict(value=bases / t_ref / 1e9, unit="Gbp/s", map_seconds=round(t_ref, 2), index_s=round(t_ref_idx, 2))
            o["vs_reference"] = round(t_ref / t_map, 1)
            o["parity"] = ("GAF byte-identical to the reference on all %d reads (%d bytes)" % (n_reads, os.path.getsize(ref))) if subprocess.call(["cmp", "-s", got, ref]) == 0 else "MISMATCH vs reference GAF"
        res[key] = o
        for f in (got, ref, log_):
            try:
                os.remove(f)
            except OSError:
                pass
    for f in (rd, pre + ".lin.fa", pre + ".gfa"):
        try:
            os.remove(f)
        except OSError:
            pass
    return res


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
    ap.add_argument("--reads", type=int, default=125000, help="reads per GPU (BASELINE configs[3]: 1M reads over 8 GPUs)")
    ap.add_argument("--genome", type=int, default=2350000000, help="backbone bp (2.35 Gbp backbone + 4 alt haplotypes = 3.02 Gbp of graph sequence)")
    ap.add_argument("--chr", type=int, default=24)
    ap.add_argument("--hap", type=int, default=5)
    ap.add_argument("--cpu-reads", type=int, default=125000, help="reads of the CPU baseline + parity sample (default: every read of one rank's share -- about 17 s of mapping on 16 cores)")
    ap.add_argument("--one-placement", action="store_true", help="skip the second placement of graph chaining (the library's own choice only)")
    ap.add_argument("--placement", choices=["auto", "device", "host"], default="auto", help="graph chaining of the headline pass: the library's own choice (auto) or forced (profiling runs)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N>1: leave every rank's GAF shard where it is (the gather to rank 0 over RCCL is ON by default)")
    ap.add_argument("--resident-steps", type=int, default=2)
    ap.add_argument("--threads", type=int, default=0, help="host threads per rank (0: min(64, usable cores / ranks))")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--gfx-activity", action="store_true", help="N=1: an UNPROFILED busy share of the GPU over extra headline steps: the driver's accumulated GFX activity counter (rocm-smi --showuse) read before and after them, scaled by the same counter over a 3 s matrix-multiply loop (block `gfx_activity`)")
    ap.add_argument("--workdir", default=None, help="keep the synthetic workload (graph, reads, graph image) in this directory and reuse it when it is already there (sweeps of several bench runs in one session)")
    ap.add_argument("--no-asm", action="store_true", help="N=1: skip the `asm` block (BASELINE configs[4] at one GPU's share: -cx asm, one ~98 Mbp contig per chromosome vs the headline's 3 Gbp graph, file -> file next to the reference, then --call)")
    ap.add_argument("--asm-genome", type=int, default=0, help="asm block on a self-made graph of this many backbone bp (10 chromosomes, 10 contigs) instead of the headline's graph (quick runs)")
    ap.add_argument("--asm-contigs", type=int, default=0, help="asm block: contigs (default: one per chromosome)")
    ap.add_argument("--no-small", action="store_true", help="N=1: skip the `linear` / `bubble50` blocks (BASELINE configs[1] / configs[2]: 100 000 reads vs the 50 Mbp linear FASTA / 3-haplotype graph)")
    ap.add_argument("--small-reads", type=int, default=100000)
    ap.add_argument("--small-genome", type=int, default=50000000)
    ap.add_argument("--no-file-out", action="store_true", help="N=1: skip the `file_out` leg (the headline's step with the GAF written to a FILE by the library's writer thread, as the CPU leg writes it)")
    ap.add_argument("--no-rank-share", action="store_true", help="N=1: skip the `rank_share` block (device placement with the process pinned to 1/8 of the usable cores)")
    ap.add_argument("--share", type=int, default=8, help="rank_share: the node's ranks the usable cores are divided among")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1: nccl (= RCCL over xGMI, the default) or gloo (host tensors: lets one GPU box run 2 ranks on the same device to test the sharded path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` launches its own ranks: one process per GPU under torch.distributed.run (the plain command ran ONE rank and said n_gpus 1)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        log("[bench] launching %d ranks: %s" % (args.gpus, " ".join(cmd)))
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or run `python bench.py --gpus %d`, which launches its own ranks)" % (args.gpus, world, args.gpus, args.gpus))
    n_gpus = world
    os.environ.setdefault("MGA_DEVICE", str(local_rank))

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(args.backend)

    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    import minigraph_amd as mga
    from minigraph_amd.dist import map_sharded
    L = mga.load()
    if L.mga_device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    ncpu = os.cpu_count() or 1
    quota = usable_cores()
    threads = args.threads or max(2, min(64, quota // world))  # [measured] more threads than usable cores only adds contention + CFS throttling

    # ---- synthetic workload (untimed): ONE graph + ONE read file for the whole job, written by rank 0 ----
    wl_tag = "G%d_c%d_H%d_n%d" % (args.genome, args.chr, args.hap, args.reads * world)
    reuse_wl = False
    if rank == 0:
        if args.workdir:
            d = os.path.join(os.path.abspath(args.workdir), wl_tag)
            os.makedirs(d, exist_ok=True)
            reuse_wl = os.path.exists(os.path.join(d, "w.done"))
        else:
            d = tempfile.mkdtemp(prefix="mga_bench_")
        t0 = time.time()
        if reuse_wl:
            err_txt = open(os.path.join(d, "w.done")).read()
        else:
            p = subprocess.run([mga.MGSIM, "-p", os.path.join(d, "w"), "-G", str(args.genome), "-c", str(args.chr), "-H", str(args.hap),
                                "-n", str(args.reads * world), "-s", "11"], stderr=subprocess.PIPE, check=True)
            err_txt = p.stderr.decode()
        m = re.search(r"graph=(\d+) bp", err_txt)
        graph_bp = int(m.group(1)) if m else int(args.genome * (1 + 0.071 * (args.hap - 1)))
        try:
            os.remove(os.path.join(d, "w.lin.fa"))
        except OSError:
            pass
        t_gen = time.time() - t0
    else:
        d, t_gen, graph_bp = None, 0.0, 0
    if dist is not None:
        box = [d]
        dist.broadcast_object_list(box, src=0)
        d = box[0]
    pre = os.path.join(d, "w")
    graph_path, reads_path = pre + ".gfa", pre + ".reads.fa"
    # graph -> index: rank 0 parses the GFA text once (gfa_read + mg_index: `index_build_s`) and writes the graph as ONE binary image; every rank then maps the image
    # and lets its GPU rebuild the minimizer table (`index_s`: what a run on an existing image pays -- SURVEY 8 f4, csrc/image.c)
    img_path = pre + ".mgi"
    t_build = t_save = 0.0
    if rank == 0 and not (reuse_wl and os.path.exists(img_path)):
        t0 = time.time()
        G0 = mga.Graph(graph_path, preset="lr", cigar=True, n_threads=threads)
        t_build = time.time() - t0
        t0 = time.time()
        G0.save_image(img_path)
        t_save = time.time() - t0
        G0.close()
        if args.workdir:
            with open(os.path.join(d, "w.done"), "w") as fo:
                fo.write(err_txt)
    if dist is not None:
        dist.barrier()
    t0 = time.time()
    G = mga.Graph(img_path, preset="lr", cigar=True, n_threads=threads, image=True)
    t_index = time.time() - t0
    log("[bench] rank %d: gen %.1fs, GFA text -> index %.1fs, image written in %.1fs, image -> index %.2fs, %d host threads" % (rank, t_gen, t_build, t_save, t_index, threads))

    last = {}

    nthr = [threads]   # (the rank_share block runs the same step with a rank's share of the threads)

    fout = [None]      # (the file_out leg: the same step with the GAF going to this file through the library's writer thread)

    def step():
        if dist is None and fout[0]:
            mga.map_files_idx(G, [reads_path], n_threads=nthr[0], out_path=fout[0])
        elif dist is None:   # (the output buffer of the previous step is handed back for reuse, like a writer thread would recycle its buffers)
            last["gaf"] = mga.map_files_idx(G, [reads_path], n_threads=nthr[0], reuse=last.get("gaf"))
        elif args.no_gather:
            last["gaf"] = mga.map_files_idx(G, [reads_path], n_threads=threads, rank=rank, world=world, reuse=last.get("gaf"))
        else:  # one input -> N GPUs -> one GAF on rank 0: size table all_gather + one RCCL gather of the bytes
            def mapper(r, w):
                m = mga.map_files_idx(G, [reads_path], n_threads=threads, rank=r, world=w, reuse=last.get("shard"))
                last["shard"] = m
                return m.view(), m.seg_len, m.cap   # (the shard's buffer is the library's: page-locked once through its registry, which unregisters before any realloc / free)
            last["gaf_bytes"] = map_sharded(mapper, dst=0, device=coll_dev, as_tensor=True)   # ONE uint8 tensor on rank 0, input order

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_warm, n_steps):
        """W untimed + K timed steps, barrier + device sync on both sides, MAX over ranks"""
        for _ in range(n_warm):
            step()
        mga.get_stats(G, reset=True)
        sync()
        cpu0, thr0 = time.process_time(), cgroup_throttled()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
        sync()
        dt_ = time.perf_counter() - t0
        cpu1, thr1 = time.process_time(), cgroup_throttled()
        st_ = mga.get_stats(G, reset=True)
        if dist is not None:
            tt = torch.tensor([dt_], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_, st_, dict(cpu_s_per_step=round((cpu1 - cpu0) / max(1, n_steps), 3), cfs_throttled_periods=thr1[0] - thr0[0], cfs_throttled_s=round((thr1[1] - thr0[1]) * 1e-6, 3))

    def last_gaf():
        """rank 0: bytes of the last step's output (enough of it for the CPU sample when it was gathered from N ranks)"""
        if rank != 0:
            return None
        if dist is None:
            return last["gaf"].bytes()
        if not args.no_gather:
            return last["gaf_bytes"][:min(int(last["gaf_bytes"].numel()), 12000 * args.cpu_reads)].cpu().numpy().tobytes()
        return None

    # Graph chaining + gap list run on the host threads or on the device (k_gchain / k_plan); the library picks by the host threads this rank has (device when <= 12).
    # The headline is the library's own choice; the OTHER placement is timed right after it (fewer steps), so that both halves of the path are inside a driver-timed,
    # parity-checked number.
    default_dev = threads <= 12 if args.placement == "auto" else args.placement == "device"
    os.environ.pop("MGA_DEV_GCHAIN", None)
    if args.placement != "auto":
        os.environ["MGA_DEV_GCHAIN"] = "1" if default_dev else "0"
    dt, st, host_main = timed(args.warmup, args.steps)
    gaf_main = last_gaf()
    gfx_block = None
    if args.gfx_activity and dist is None:
        import re as _re

        def gfx_acc():
            try:
                o_ = subprocess.run(["rocm-smi", "--showuse"], capture_output=True, text=True, timeout=30).stdout
                m_ = _re.search(r"GFX Activity:\s*(\d+)", o_)
                return int(m_.group(1)) if m_ else None
            except Exception:
                return None
        try:
            a0, t0_ = gfx_acc(), time.perf_counter()
            x_ = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
            while time.perf_counter() - t0_ < 3.0:
                for _ in range(20):
                    x_ = (x_ @ x_).clamp_(-1, 1)
                torch.cuda.synchronize()
            a1, t1_ = gfx_acc(), time.perf_counter()
            del x_
            k_g = max(10, args.steps)
            b0, u0 = gfx_acc(), time.perf_counter()
            dt_g, st_g, _h = timed(0, k_g)
            b1, u1 = gfx_acc(), time.perf_counter()
            if None not in (a0, a1, b0, b1) and a1 > a0:
                per_s_busy = (a1 - a0) / (t1_ - t0_)
                gfx_block = dict(busy_share=round((b1 - b0) / (u1 - u0) / per_s_busy, 3), steps=k_g, value=st_g["n_bases"] / dt_g / 1e9, ms_per_step=dt_g / k_g * 1e3,
                                 counter_per_s_when_busy=round(per_s_busy, 1), counter_per_s_over_the_steps=round((b1 - b0) / (u1 - u0), 1),
                                 note="UNPROFILED: the driver's accumulated GFX activity counter (rocm-smi --showuse) over %d headline steps against the same counter over 3 s of back-to-back bf16 matrix products" % k_g)
        except Exception as e_:
            gfx_block = dict(error=str(e_))
    file_out = None
    if dist is None and not args.no_file_out:   # SURVEY 8d asks for the same interval on both sides: the reference's step 2 writes its GAF to a file (gmap.c:119-139), so does this leg
        fout[0] = os.path.join(d, "gpu.gaf")
        k_fo = max(1, min(args.steps, 3))
        dt_f, st_f, host_f = timed(1, k_fo)
        file_out = dict(value=st_f["n_bases"] / dt_f / 1e9, unit="Gbp/s", ms_per_step=dt_f / k_fo * 1e3, steps=k_fo, warmup=1, path=fout[0], **host_f,
                        note="the headline's step with the GAF written to a file by the library's writer thread (fwrite per mini-batch, overlapped with the mapping of the next ones), as the CPU leg's stdout goes to a file")
        fout[0] = None
    os.environ.pop("MGA_DEV_GCHAIN", None)
    other = None
    if not args.one_placement:
        os.environ["MGA_DEV_GCHAIN"] = "0" if default_dev else "1"
        k_other = max(1, min(args.steps, 3))
        dt_o, st_o, host_o = timed(1, k_other)
        other = dict(dt=dt_o, steps=k_other, st=st_o, host=host_o, gaf=last_gaf())
        os.environ.pop("MGA_DEV_GCHAIN", None)
    # ---- rank_share (N = 1): what ONE rank of an 8-GPU node gets of this box -- the device placement with 1/8 of the usable cores: every thread of the process (the HIP
    #      runtime's included) pinned to that many CPUs, as many host threads.  Timed by the same function as the headline. ----
    share = None
    if dist is None and not args.no_rank_share and not args.one_placement:
        cores_share = max(1, quota // max(1, args.share))
        allowed, before = sorted(os.sched_getaffinity(0)), {}
        try:
            pin = set(allowed[:cores_share])
            tids = [int(t) for t in os.listdir("/proc/self/task")]
            for t in tids:
                try:
                    before[t] = os.sched_getaffinity(t)
                    os.sched_setaffinity(t, pin)
                except OSError:
                    pass
            L.mga_idx_stream_close.argtypes = [ctypes.c_void_p]
            L.mga_idx_stream_close(G.gi)          # the index's pipeline threads are created anew, inside the pinned set
            os.environ["MGA_DEV_GCHAIN"] = "1"
            nthr[0] = max(2, cores_share)
            k_share = max(1, min(args.steps, 3))
            dt_s, st_s, host_s = timed(1, k_share)
            share = dict(value=st_s["n_bases"] / dt_s / 1e9, unit="Gbp/s", ms_per_step=dt_s / k_share * 1e3, steps=k_share, warmup=1, cores_pinned=cores_share, host_threads=nthr[0],
                         note="device placement of graph chaining + gap list; the whole process pinned to %d of the %d usable cores (1/%d: a rank's share of this box on a node of %d GPUs)"
                              % (cores_share, quota, args.share, args.share), gaf=last_gaf(), **host_s)
        finally:
            nthr[0] = threads
            os.environ.pop("MGA_DEV_GCHAIN", None)
            for t, m in before.items():
                try:
                    os.sched_setaffinity(t, m)
                except OSError:
                    pass
            for t in os.listdir("/proc/self/task"):   # threads born while pinned
                try:
                    os.sched_setaffinity(int(t), set(allowed))
                except OSError:
                    pass
            L.mga_idx_stream_close(G.gi)
    n_reads_rank, n_bases_rank = st["n_reads"] // max(1, args.steps), st["n_bases"] // max(1, args.steps)
    if dist is not None:
        tb = torch.tensor([n_bases_rank, n_reads_rank] + [st[k] for k in ("n_mz", "n_hit", "wfa_t_bases", "wfa_q_bases", "gaf_bytes")], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(tb)
        total_bases, total_reads = int(tb[0].item()), int(tb[1].item())
        agg = dict(zip(("n_mz", "n_hit", "wfa_t_bases", "wfa_q_bases", "gaf_bytes"), (int(x) for x in tb[2:].tolist())))
    else:
        total_bases, total_reads = n_bases_rank, n_reads_rank
        agg = {k: st[k] for k in ("n_mz", "n_hit", "wfa_t_bases", "wfa_q_bases", "gaf_bytes")}

    # ---- second key: the same reads resident in HBM (no parse, no upload), rank 0 only, untimed extras ----
    resident = isolated = isolated_other = None
    if rank == 0:
        gaf_first = gaf_main
        if dist is None and last.get("gaf") is not None:
            last["gaf"].free()
        if args.resident_steps > 0:
            R = mga.Reads(reads_path, max_reads=args.reads)
            mga.map_reads(G, R, n_threads=threads, copy=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.resident_steps):
                mga.map_reads(G, R, n_threads=threads, copy=False)
            torch.cuda.synchronize()
            dtr = (time.perf_counter() - t0) / args.resident_steps
            resident = dict(value=R.bases / dtr / 1e9, unit="Gbp/s", ms_per_step=dtr * 1e3, reads=R.n,
                            note="one GPU, reads already in HBM (mga_reads_load), GAF text into the library's buffer: no FASTA parse, no upload")
            # ---- isolated passes: ONE chunk in flight (MGA_PIPE=1), per-kernel HIP-event times on the launch stream do not overlap; one per placement ----
            L.mga_idx_stream_close.argtypes = [ctypes.c_void_p]
            L.mga_wfa_ladder_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
            import numpy as np

            def isolated_pass(dev):
                os.environ["MGA_PIPE"] = "1"
                os.environ["MGA_WFA_SIDE"] = "0"    # (the register tiers' own problems normally start early on a side stream: serialised here, so that launch durations do not overlap)
                os.environ["MGA_DEV_GCHAIN"] = "1" if dev else "0"
                L.mga_idx_stream_close(G.gi)        # the index's chunk pipeline is rebuilt with one pipeline thread for this pass
                mga.get_stats(G, reset=True)
                mga.prof_enable(True)
                mga.prof_get(reset=True)
                _z = (ctypes.c_int64 * 32)()
                L.mga_wfa_ladder_stats(_z, ctypes.byref(_z, 128), 1)
                t0_ = time.perf_counter()
                m = mga.map_files_idx(G, [reads_path], n_threads=threads, rank=0, world=world)
                torch.cuda.synchronize()
                dti = time.perf_counter() - t0_
                m.free()
                del os.environ["MGA_PIPE"], os.environ["MGA_DEV_GCHAIN"], os.environ["MGA_WFA_SIDE"]
                L.mga_idx_stream_close(G.gi)
                prof_, sti_ = mga.prof_get(), mga.get_stats(G)
                ln, lu = np.zeros(16, dtype=np.int64), np.zeros(16, dtype=np.int64)
                L.mga_wfa_ladder_stats(ln.ctypes.data, lu.ctypes.data, 0)
                mga.prof_enable(False)
                return dict(ms=dti * 1e3, prof=prof_, st=sti_, ladder=dict(rungs=["W16x4", "W32x2", "W64", "W128", "W192", "W256", "R512", "R1024", "R2048", "H4096", "H32768"],
                                                                      run=[int(x) for x in ln[:11]], arrived_from_below=[int(x) for x in lu[:11]]))
            isolated = isolated_pass(default_dev)
            if not args.one_placement:
                isolated_other = isolated_pass(not default_dev)
            R.close()
    if dist is not None:
        dist.barrier()

    if rank == 0:
        value = total_bases * args.steps / dt / 1e9
        # algorithmic bytes of the whole path per read = L + 16 n_mz + 8 n_hit + 32 n_hit + T + Q + O, from this run's counters
        b_alg = (total_bases * args.steps + 16 * agg["n_mz"] + 40 * agg["n_hit"] + agg["wfa_t_bases"] + agg["wfa_q_bases"] + agg["gaf_bytes"])
        path_ach = b_alg / dt / 1e9
        roof_path = dict(bound="hbm", scope="whole path (SURVEY 8d: B_alg = L + 16 n_mz + 8 n_hit + 32 n_hit + T + Q + O, summed over this run's reads)",
                         achieved=path_ach, peak=HBM_PEAK_GBS * n_gpus, unit="GB/s", frac=path_ach / (HBM_PEAK_GBS * n_gpus),
                         alg_bytes_per_read=b_alg / max(1, total_reads * args.steps))
        roof = None
        kernels_ms = {}
        if isolated:
            prof, sti = isolated["prof"], isolated["st"]
            kernels_ms = {k: round(v[0], 3) for k, v in prof.items()}
            wfa = [k for k in prof if k.startswith("k_wfa")]
            fam = {"k_wfa": sum(prof[k][0] for k in wfa), "k_sketch": prof["k_sketch"][0],
                   "k_seed": prof["k_seed_count"][0] + prof["k_seed_fill"][0], "k_lchain": prof["k_lchain"][0], "k_text": prof["k_text"][0]}
            launches = {"k_wfa": sum(prof[k][1] for k in wfa), "k_sketch": prof["k_sketch"][1],
                        "k_seed": prof["k_seed_count"][1] + prof["k_seed_fill"][1], "k_lchain": prof["k_lchain"][1], "k_text": prof["k_text"][1]}
            alg = {  # algorithmic bytes of each kernel family over the isolated pass
                "k_wfa": sti["wfa_t_bases"] + sti["wfa_q_bases"],
                "k_sketch": sti["n_bases"] + 16 * sti["n_mz"],
                "k_seed": 16 * sti["n_probe"] + 8 * sti["n_hit"] + 16 * sti["n_hit"],
                "k_lchain": 32 * sti["n_hit"],
                "k_text": sti["gaf_bytes"],
            }
            dom = max(fam, key=lambda k: fam[k])
            # HBM traffic of the dominant family per launch: from the committed PMC passes of this command (profiles/*_pmc.json); bench.py
            # cannot run rocprofv3 around itself, so the figure is labelled with its source and is NOT measured in this run
            traffic, traffic_src = None, None
            try:
                import glob
                pf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]*_pmc.json")))[-1]
                pk = json.load(open(pf))["kernels"]
                sel = [v for k, v in pk.items() if (k == "k_wfa" or k.startswith("k_wfa_r<") or k.startswith("k_wfa_fw<") or k.startswith("k_wfa_fwp<") or k.startswith("k_wfa_tb") if dom == "k_wfa" else k.startswith(dom))]
                nl = sum(v.get("launches_fetch", 0) for v in sel)
                if nl:
                    traffic = sum(v.get("fetch_kb", 0) + v.get("write_kb", 0) for v in sel) * 1024.0 / nl
                    import hashlib
                    h = hashlib.sha1()   # the profile names the WFA sources it was taken from (prof_summary.py --pmc-json); a profile of other sources says so
                    for f in sorted(glob.glob(os.path.join(ROOT, "minigraph_amd", "csrc", "k_wfa*.hip")) + [os.path.join(ROOT, "minigraph_amd", "csrc", "wfa_window.h")]):
                        h.update(open(f, "rb").read())
                    stale = json.load(open(pf)).get("wfa_src_sha1") != h.hexdigest()
                    traffic_src = os.path.relpath(pf, ROOT) + " (committed rocprofv3 --pmc passes of this command, not measured in this run%s)" % ("; STALE: taken from other WFA kernel sources than this tree's" if stale else "")
                    all_rec = json.load(open(pf)).get("kernel_src_sha1")   # (round 5) every device source, whichever family is priced
                    if all_rec is not None and all_rec != kernel_src_sha1():
                        traffic_src += "; STALE: the tree's device sources differ from the ones the passes ran"
            except Exception:
                pass
            # vector-ALU utilisation per rung of the WFA ladder (and the other big kernels): vector instructions per launch -- SQ_INSTS_VALU of the committed counter passes of this
            # command (profiles/*_sq_counters.txt, tools/prof_all.sh: bench.py cannot collect counters around itself) -- x 4.15 cycles of a SIMD per wave64 integer instruction
            # ([measured] profiles/r03_valu_rate.txt) over the SIMD cycles of the launch as timed HERE (HIP events of the isolated pass): the share of the chip's vector issue
            # slots the kernel uses while it runs
            valu_busy, valu_src = None, None
            try:
                import glob
                sf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[4-9]*_sq_counters.txt")))[-1]
                sq = {}
                for ln in open(sf):
                    f = ln.split()
                    if ln.startswith("#") or len(f) < 14 or f[0] == "kernel":
                        continue
                    try:
                        sq[" ".join(f[:-13])] = dict(calls=int(f[-13]), waves=float(f[-12]), valu_per_wave=float(f[-11]), valu_share=float(f[-8]) if f[-8] != "-" else None,
                                                     wait_share=float(f[-7]) if f[-7] != "-" else None)
                    except ValueError:
                        pass
                # (round 5: the 128 / 192 / 256 rungs run the packed kernel, MGA_WFA_PACKED=7)
                names = {"k_wfa_w[16x4]": "k_wfa_fw<16, 1, 128>", "k_wfa_w[32x2]": "k_wfa_fw<32, 1, 192>", "k_wfa_w[64]": "k_wfa_fw<64, 1, 256>", "k_wfa_w[128]": "k_wfa_fwp<128, 384>",
                         "k_wfa_w[192]": "k_wfa_fwp<192, 384>", "k_wfa_w[256]": "k_wfa_fwp<256, 512>", "k_wfa_r[512]": "k_wfa_r<4, 2, 1024, 1024, 8192, true>", "k_wfa_tb": "k_wfa_tb",
                         "k_lchain": "k_lchain<6>", "k_sketch": "k_sketch<128>", "k_text": "k_text_w", "k_seed_fill": "k_seed_fill", "k_seed_count": "k_seed_count", "k_gaf": "k_gaf<true>"}
                for kn_, alt in (("k_lchain", "k_lchain"), ("k_sketch", "k_sketch"), ("k_text", "k_text<64>")):  # counter files of rounds 4-5 (before the kernels were templates)
                    if names[kn_] not in sq and alt in sq:
                        names[kn_] = alt
                # the counter passes ran `bench.py --steps S --warmup W --one-placement` over the SAME reads as this run's isolated pass, so a kernel's instructions per pass =
                # its total / the passes the file covers; launches are not compared one to one (the chunking of a pass may differ).  k_sketch is left out: its
                # counters include the index build's launches over the graph
                # (round 5, VERDICT r4 weak 1: the pass count comes from the counter file itself -- k_lchain is one wavefront per read, so its waves / the reads of a pass IS the
                # number of passes the counters cover; taking it from --steps / --warmup + "the isolated pass" counted a pass the profiled command never ran)
                n_pass = int(round(sq[names["k_lchain"]]["waves"] / float(args.reads))) if names["k_lchain"] in sq and args.reads > 0 else 0
                if n_pass < 1:
                    raise ValueError("no pass count")
                names.pop("k_sketch")
                for kn_, alt in (("k_wfa_w[128]", "k_wfa_fw<64, 2, 384>"), ("k_wfa_w[192]", "k_wfa_fw<64, 3, 384>"), ("k_wfa_w[256]", "k_wfa_fw<64, 4, 512>")):  # a counter file taken with MGA_WFA_PACKED=0 / before round 5
                    if names[kn_] not in sq and alt in sq:
                        names[kn_] = alt
                valu_busy = {}
                for kn, sn in names.items():
                    q, pr_ = sq.get(sn), prof.get(kn)
                    if not q or not pr_ or pr_[0] <= 0 or pr_[1] <= 0 or q["calls"] <= 0:
                        continue
                    inst_per_pass = q["valu_per_wave"] * q["waves"] / n_pass
                    valu_busy[kn] = dict(valu_busy=round(inst_per_pass * 4.15 / (256 * 4 * pr_[0] * 1e-3 * 2.4e9), 3), valu_instr_per_pass=round(inst_per_pass), ms_per_pass=round(pr_[0], 2),
                                         share_of_a_waves_cycles=dict(valu=q["valu_share"], waiting=q["wait_share"]))
                sq_rec = None
                for ln in open(sf):
                    if ln.startswith("# kernel_src_sha1:"):
                        sq_rec = ln.split(":", 1)[1].strip()
                        break
                sq_lbl = "" if sq_rec == kernel_src_sha1() else ("; no source hash recorded in the file" if sq_rec is None else "; STALE: the tree's device sources differ from the ones the passes ran")
                valu_src = ("VALU instructions per pass of 125000 reads: " + os.path.relpath(sf, ROOT) + sq_lbl + " (committed rocprofv3 --pmc SQ_INSTS_VALU / SQ_WAVES passes of this command, %d passes); "
                            "kernel time per pass: this run's isolated pass; 4.15 SIMD cycles per wave64 integer instruction [measured], 1024 SIMDs at the nominal 2.4 GHz "
                            "(a lower sustained clock raises the true figure)" % n_pass)
            except Exception:
                valu_busy = None
            ach = alg[dom] / (fam[dom] * 1e-3) / 1e9 if fam[dom] > 0 else 0.0
            roof = dict(bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                        launches=launches[dom], avg_launch_ms=fam[dom] / max(1, launches[dom]), alg_bytes_per_launch=alg[dom] / max(1, launches[dom]),
                        family_ms_per_pass=fam[dom], pass_ms=isolated["ms"],
                        note="dominant kernel family by HIP-event time in the ISOLATED pass (one chunk in flight: launch durations do not overlap, sum <= pass_ms); "
                             "the family is bound by vector-instruction ISSUE, not by HBM: int_issue prices it against the measured integer issue rate of a SIMD (profiles/r03_valu_rate.txt)",
                        families={k: dict(ms=round(fam[k], 2), launches=launches[k], alg_GBps=round(alg[k] / max(fam[k], 1e-9) / 1e6, 2)) for k in fam},
                        valu_busy=valu_busy, valu_busy_source=valu_src)
            if traffic is not None and launches[dom] > 0 and alg[dom] > 0:
                roof["traffic_ratio"] = round(traffic / (alg[dom] / launches[dom]), 2)  # HBM bytes moved per algorithmic byte of the dominant family (1 = nothing re-read or spilled)
            # per kernel family: HBM bytes of the committed counter passes per pass against the family's algorithmic bytes, and what binds it (vector issue when the kernel uses
            # more than half of the chip's measured issue rate while it runs, else the latency of its dependent accesses: DESIGN.md 4)
            try:
                pj = json.load(open(pf))
                n_pass_pmc = max(1, int(round(sum(v.get("launches_fetch", 0) for k, v in pj["kernels"].items() if k.startswith("k_lchain")) / max(1, launches["k_lchain"]))))
                fam_of = lambda k: ("k_wfa" if (k == "k_wfa" or k.startswith("k_wfa_r<") or k.startswith("k_wfa_fw") or k.startswith("k_wfa_tb")) else "k_seed" if k.startswith("k_seed") else
                                    "k_text" if k.startswith("k_text") else "k_sketch" if k.startswith("k_sketch<") or k == "k_sketch" else "k_lchain" if k.startswith("k_lchain") else None)
                tr = {}
                for k, v in pj["kernels"].items():
                    f_ = fam_of(k)
                    if f_ and f_ != "k_sketch":   # (k_sketch's counters include the index build over the graph)
                        tr[f_] = tr.get(f_, 0.0) + (v.get("fetch_kb", 0) + v.get("write_kb", 0)) * 1024.0 / n_pass_pmc
                busy_of = {"k_wfa": ["k_wfa_w[64]", "k_wfa_w[128]"], "k_lchain": ["k_lchain"], "k_text": ["k_text"], "k_seed": ["k_seed_fill"]}
                per = {}
                for f_, b_ in tr.items():
                    vb = [valu_busy[x]["valu_busy"] for x in busy_of.get(f_, []) if valu_busy and x in valu_busy]
                    per[f_] = dict(hbm_bytes_per_pass=round(b_), alg_bytes_per_pass=round(alg[f_]), traffic_ratio=round(b_ / alg[f_], 2) if alg[f_] > 0 else None,
                                   hbm_frac_of_peak=round(b_ / (fam[f_] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if fam[f_] > 0 else None,
                                   bound=("vector issue" if vb and max(vb) >= 0.5 else "latency of dependent accesses / occupancy"))
                roof["per_family"] = per
                roof["per_family_source"] = traffic_src
            except Exception:
                pass
            # integer-issue roofline of the WFA family: wavefront cells (the REFERENCE's band: what miniwfa computes for the same gaps) per second against what the vector
            # ALUs can issue.  [measured, minigraph_amd/tools/valu_rate.hip -> profiles/r03_valu_rate.txt] a gfx950 SIMD issues one wave64 v_max_i32 / v_add_u32 /
            # DPP move every 4.15 cycles; a 64-cell slot step of the windowed kernel is ~110 such instructions (ISA count incl. one mask-window extension)
            cells = sti["wfa_cells"]
            if fam["k_wfa"] > 0 and cells > 0:
                clk, simd, cyc_per_instr, instr_per_slot_step = 2.4e9, 256 * 4, 4.15, 110.0
                peak_cells = simd * clk / cyc_per_instr * 64.0 / instr_per_slot_step
                roof["int_issue"] = dict(bound="valu-issue", achieved=cells / (fam["k_wfa"] * 1e-3), peak=peak_cells, unit="reference-band wavefront cells/s",
                                         frac=cells / (fam["k_wfa"] * 1e-3) / peak_cells,
                                         note="peak = 256 CU x 4 SIMD x 2.4 GHz / 4.15 cycles per wave64 integer instruction [measured] x 64 cells / 110 instructions per slot step; achieved counts the cells of "
                                              "the reference's band (miniwfa.c:421 n_iter) -- the windowed tiers compute about 2.4x fewer to the same alignment, which is why frac is not a utilisation")
        _gfx_block = gfx_block
        res = dict(metric=METRIC, value=value, unit="Gbp/s",
                   n_gpus=n_gpus, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="u8/int32", data="synthetic",
                   config=dict(workload="configs[3] shape: %d x 10kb synthetic ONT reads per GPU (%d in all, ONE file) vs %.2f Gbp %d-haplotype bubble graph in %d chromosomes, -cx lr -c"
                               % (args.reads, total_reads, graph_bp / 1e9, args.hap, args.chr),
                               interval="FASTA file -> parse -> H2D -> map -> GAF text in one memory buffer per rank%s (reference: worker_pipeline - mg_opt_update); graph load + index build excluded (index_s)"
                               % ((" -> %s gather to rank 0 -> one GAF in input order" % ("RCCL" if args.backend == "nccl" else args.backend)) if (n_gpus > 1 and not args.no_gather) else ""),
                               reads_per_gpu=args.reads, read_len=10000, err=0.1,
                               sharding="1 input file cut into %d contiguous byte ranges at record starts, index replicated%s" % (n_gpus, (", GAF gathered to rank 0 over %s" % ("RCCL" if args.backend == "nccl" else args.backend)) if (n_gpus > 1 and not args.no_gather) else ""),
                               host_threads_per_rank=threads),
                   roofline=roof if roof else roof_path, roofline_path=roof_path,
                   resident=resident,
                   kernels_ms_isolated=kernels_ms,
                   wfa_ladder_isolated=(isolated or {}).get("ladder"),
                   graph_chaining=dict(default=("device (k_gchain + k_plan)" if default_dev else "host threads for %d %% of the chunks, device for the rest (MGA_DEV_GCHAIN_PCT, default 25)" % (100 - int(os.environ.get("MGA_DEV_GCHAIN_PCT", "25")))) + ": %d host threads per rank, device for every chunk when <= 12" % threads + ("" if args.placement == "auto" else " (FORCED by --placement %s)" % args.placement)),
                   per_read=dict(n_mz=agg["n_mz"] / max(1, total_reads * args.steps), n_hit=agg["n_hit"] / max(1, total_reads * args.steps),
                                 n_wfa=st["n_wfa"] / max(1, st["n_reads"]), wfa_cells=st["wfa_cells"] / max(1, st["n_reads"]),
                                 gaf_bytes=agg["gaf_bytes"] / max(1, total_reads * args.steps)),
                   index_s=round(t_index, 2), index_build_s=round(t_build, 2),
                   index_note="index_s: graph image (mga_graph_image_save) mapped + minimizer table rebuilt on the device, per rank; index_build_s: gfa_read + mg_index from the GFA text (rank 0, once)",
                   host=dict(logical_cpus=ncpu, usable_cores=quota, **host_main))
        if other is not None:   # the other placement of graph chaining + gap list, timed the same way (fewer steps)
            key = "host_placement" if default_dev else "device_placement"
            res[key] = dict(value=total_bases * other["steps"] / other["dt"] / 1e9, unit="Gbp/s", ms_per_step=other["dt"] / other["steps"] * 1e3, steps=other["steps"], warmup=1,
                            forced_by="MGA_DEV_GCHAIN=%d" % (0 if default_dev else 1), **other["host"])
            if isolated_other:
                res[key]["kernels_ms_isolated"] = {k: round(v[0], 3) for k, v in isolated_other["prof"].items() if v[0] > 0}
        share_gaf = None
        if share is not None:
            share_gaf = share.pop("gaf")
            dp = res.get("device_placement", {}).get("value") if not default_dev else value
            share["vs_device_placement"] = round(share["value"] / dp, 3) if dp else None
            res["rank_share"] = share
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "minigraph")
        if not args.no_cpu and os.path.exists(ref_bin):
            try:
                cb, cpu_gaf, n_cpu = cpu_baseline(ref_bin, graph_path, reads_path, args.cpu_reads, d, min(ncpu, 2 * quota), quota)
                res["cpu_baseline"] = cb
                want = open(cpu_gaf, "rb").read()

                def same(got):
                    return len(got) >= len(want) and got[:len(want)] == want and (len(got) == len(want) or got[len(want) - 1:len(want)] == b"\n")
                if gaf_first is not None:
                    res["parity"] = ("GAF byte-identical to the reference on ALL %d reads of the CPU sample (%d bytes)" % (n_cpu, len(want))) if same(gaf_first) else "MISMATCH vs reference GAF"
                if share_gaf is not None:
                    res["rank_share"]["parity"] = ("GAF byte-identical to the reference on ALL %d reads of the CPU sample (%d bytes)" % (n_cpu, len(want))) if same(share_gaf) else "MISMATCH vs reference GAF"
                if other is not None and other["gaf"] is not None:
                    res["host_placement" if default_dev else "device_placement"]["parity"] = \
                        ("GAF byte-identical to the reference on ALL %d reads of the CPU sample (%d bytes)" % (n_cpu, len(want))) if same(other["gaf"]) else "MISMATCH vs reference GAF"
                if file_out is not None:
                    fo_path = file_out.pop("path")
                    if n_cpu >= args.reads:   # whole files, compared as files
                        file_out["parity"] = ("GAF file byte-identical to the reference's (%d bytes)" % len(want)) if subprocess.call(["cmp", "-s", fo_path, cpu_gaf]) == 0 else "MISMATCH vs reference GAF file"
                    else:
                        with open(fo_path, "rb") as fi:
                            file_out["parity"] = ("GAF file byte-identical to the reference on the %d reads of the CPU sample" % n_cpu) if same(fi.read(len(want) + 1)) else "MISMATCH vs reference GAF file"
                del want
            except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
                res["cpu_baseline"] = dict(value=None, unit="Gbp/s", cores=quota, kind="reference", sample="failed: %r" % (e,))
        else:
            res["cpu_baseline"] = dict(value=None, unit="Gbp/s", cores=quota, kind="reference",
                                       sample="not run (--no-cpu, or oracle/_ref/minigraph absent)")
        if file_out is not None:
            file_out.pop("path", None)
            res["file_out"] = file_out
        if dist is None and not (args.no_asm and args.no_small):
            # the other BASELINE configs, each at its size, each compared with the reference: the headline's index and buffers are released first (host and HBM are free for
            # the jobs below, which load their own graphs like any file -> file job does)
            gaf_first = share_gaf = gaf_main = None
            if other is not None:
                other["gaf"] = None
            last.clear()
            G.close()
            G = None
            import gc
            gc.collect()
            if not args.workdir and not args.keep:   # the headline's files are not needed any more (the asm block maps against the graph's GFA text only)
                for f in (reads_path, img_path, os.path.join(d, "cpu.gaf"), os.path.join(d, "gpu.gaf"), os.path.join(d, "cpu_sample.fa")):
                    try:
                        os.remove(f)
                    except OSError:
                        pass
        if dist is None and not args.no_small:   # ---- BASELINE configs[1], configs[2] at their sizes ----
            try:
                res.update(small_configs_block(mga, d, threads, None if args.no_cpu else ref_bin, args.small_reads, args.small_genome))
            except Exception as e:
                res["linear"] = dict(error=repr(e))
        if dist is None and not args.no_asm:   # ---- BASELINE configs[4] at one GPU's share: -cx asm on chromosome-scale contigs vs the 3 Gbp graph, file -> file, then --call ----
            try:
                res["asm"] = asm_block(mga, d, dict(genome=args.genome, chr=args.chr, hap=args.hap), graph_path, threads, None if args.no_cpu else ref_bin,
                                       os.path.join(ROOT, "oracle", "_ref", "minigraph_dropin"), n_contig=args.asm_contigs, small_genome=args.asm_genome)
            except Exception as e:
                res["asm"] = dict(error=repr(e))
        if _gfx_block is not None:
            res["gfx_activity"] = _gfx_block
        print(json.dumps(res), flush=True)
    if G is not None:
        G.close()
    if dist is not None:
        dist.barrier()
    if rank == 0 and not args.keep and not args.workdir:
        shutil.rmtree(d, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
