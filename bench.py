#!/usr/bin/env python3
"""bench.py -- REAL read-matching hot path on MI355X.

One step = one pass of the hot path (pack -> signatures -> bucket lookup -> seed
filter -> verify -> score -> best/unique fold) over one batch of synthetic reads
that is already resident in HBM.  Workload = BASELINE.json configs[1] (C2):
matchUnique, 50M synthetic 100 bp FASTQ reads vs a 3 Gbp synthetic genome,
k=3 (seed k<=2), scores on.  With --gpus N every rank holds the whole index
(replicated) and its own shard of reads (weak scaling, C4 = 8 x 50M); the only
data-path communication is one RCCL gather of the per-read records (matchUnique)
or of the variable-length hit lists (--mode all) to rank 0.

  python bench.py --gpus N --steps K --warmup W

--gpus N > 1 without a torch.distributed environment starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` as a fresh child
process, before anything touches the GPU) and forwards the child's JSON line; inside
such a launch (RANK/WORLD_SIZE set, by this launcher or by the driver's) it runs as one rank.

Prints ONE JSON line on rank 0.  At N=1 the line also carries, under "extra": the same
step on reads in shuffled order, C3 (matchAll, k=2, same index) and C5 (150 bp, 64-bit
signatures, k=5, index rebuilt), each with its own ms/step, roofline and a check against
the oracle, and the host-inclusive (pinned, double-buffered, packed) rate of C2.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "aligned reads/sec (100 bp, k≤3) at 1/2/4/8 MI355X; achieved HBM GB/s vs peak"
TABLE_KINDS = ("starts", "digest", "fingerprint", "rows")
NO_SCORE = -3.4028234663852886e38   # UniqueMatchInfo<true>(): score(-FLT_MAX), UniqueMatchInfo.hpp:191


def algorithmic_bytes(c, patl, seedl, scores, hit_bytes_out=0):
    """A = R*B_io + L*8 + P*w + C*(8+w) + S*(3*18+8+8*W_rest) + H*(18+8*W_score),
    evaluated with the kernel's own work counters (P = index entries examined); matchAll adds the
    hit records it writes."""
    w = 4 if seedl <= 32 else 8
    w_rest = -(-(patl - seedl) // 32) + 1
    w_score = (-(-patl // 32) + 1) if scores else 0
    b_io = -(-patl // 4) + (patl if scores else 0) + 2 * (8 + (4 if scores else 0))
    return (c["reads"] * b_io + c["lookups"] * 8 + c["probes"] * w + c["candidates"] * (8 + w) +
            c["seedpass"] * (3 * 18 + 8 + 8 * w_rest) + c["hits"] * (18 + 8 * w_score) + hit_bytes_out)


def kernel_source_hash(git_rev=None):
    """sha256 over the sources of the match kernel with comments and white space taken out: profiles/traffic.json records
    the hash it was measured with, and a figure measured on another kernel is not reported.  git_rev: of the files as that
    commit holds them (collect_profiles.py, for passes that ran on a commit the working tree has moved on from)."""
    import re
    h = hashlib.sha256()
    for f in ("match_kernel.hip", "match_common.h", "kernel_common.h", "real_hip_internal.h"):
        if git_rev:
            src = subprocess.run(["git", "-C", ROOT, "show", "%s:real_amd/csrc/%s" % (git_rev, f)], stdout=subprocess.PIPE, check=True).stdout.decode("utf-8")
        else:
            with open(os.path.join(ROOT, "real_amd", "csrc", f), "r", encoding="utf-8") as fh:
                src = fh.read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)      # block comments
        src = re.sub(r"//[^\n]*", " ", src)                    # line comments (no string of these files holds "//")
        h.update(" ".join(src.split()).encode())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------
# launcher: --gpus N starts N ranks (one process per GPU) unless we already are one of them
# ---------------------------------------------------------------------------------------------
def self_launch(n_ranks):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same args>` as a fresh child
    (this process has not touched the GPU -- torch is not even imported yet), let it print the one JSON line, and
    return its exit status."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_check(args):
    """--launch-check: the ranks of this launch meet over gloo on the CPU (no GPU call), count each other with one
    all_reduce and rank 0 prints what it saw.  The CPU test of the launcher."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    seen = 1
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": seen, "requested": args.gpus,
                          "world_size_env": world, "backend": "gloo" if world > 1 else None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
# synthetic workload (seeded generators on the device: bench_support/synth_kernels.hip)
# ---------------------------------------------------------------------------------------------
def _synth():
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "real_amd", "libreal_synth.so"))
    L.real_synth_genome.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.real_synth_positions.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.real_synth_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_double, C.c_uint64,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def gen_genome(torch, G, seed, device):
    """randstr.cpp:27-53: i.i.d. uniform ACGT, generated on the device."""
    sym = torch.empty(G, dtype=torch.uint8, device=device)
    torch.cuda.synchronize()
    rc = _synth().real_synth_genome(sym.data_ptr(), G, seed)
    assert rc == 0, "synth_genome failed: hip error %d" % rc
    return sym


def gen_reads(torch, sym, n_reads, patl, errprob, seed, device, shuffle=False):
    """genpat's distribution (genpat.cpp:96-157) on the device: sorted uniform start positions (genpat.cpp:99),
    strand flip p=0.5, per-base substitution to a different base with probability errprob, quality 35 ('D'-33)
    unchanged / 9 ('*'-33) mutated.  shuffle: the same reads in random order (a sequencer does not sort)."""
    L = _synth()
    n = sym.shape[0]
    pos = torch.empty(n_reads, dtype=torch.int64, device=device)
    torch.cuda.synchronize()
    rc = L.real_synth_positions(pos.data_ptr(), n_reads, n - patl + 1, seed)
    assert rc == 0
    if not shuffle:
        pos, _ = torch.sort(pos)          # (unsorted uniform samples ARE a uniformly shuffled order)
    bases = torch.empty(n_reads * patl, dtype=torch.uint8, device=device)
    qual = torch.empty(n_reads * patl, dtype=torch.uint8, device=device)
    inv = torch.empty(n_reads, dtype=torch.uint8, device=device)
    torch.cuda.synchronize()
    rc = L.real_synth_reads(sym.data_ptr(), pos.data_ptr(), n_reads, patl, errprob, seed, bases.data_ptr(), qual.data_ptr(), inv.data_ptr())
    assert rc == 0, "synth_reads failed: hip error %d" % rc
    return bases, qual, pos, inv


def pack_bases(torch, bases, n_reads, patl):
    """one symbol per byte -> 2 bits per base, four per byte, MSB first (real_hip_batch.packed; the reads of the workload
    hold no N).  What the matcher reads from HBM per 100 bp read drops from 100 to 25 bytes of bases."""
    assert (n_reads * patl) % 4 == 0          # (a read may start inside a byte: 150 bp reads alternate)
    q4 = bases.view(n_reads * patl // 4, 4)
    return ((q4[:, 0] << 6) | (q4[:, 1] << 4) | (q4[:, 2] << 2) | q4[:, 3]).to(torch.uint8)


def strided_sample(torch, bases, qual, n_reads, patl, k):
    """k reads of the batch at a constant stride, as host arrays (no gather kernels on >2^31-element tensors)"""
    stride = max(1, n_reads // max(k, 1))
    b = bases.view(n_reads, patl)[::stride][:k].contiguous().cpu().numpy().reshape(-1)
    q = qual.view(n_reads, patl)[::stride][:k].contiguous().cpu().numpy().reshape(-1)
    k = b.shape[0] // patl
    off = (np.arange(k + 1, dtype=np.uint64) * np.uint64(patl))
    return b, q, off, stride, k


def host_cpu_info():
    """what the box's host side is: CPU model, logical CPUs of the machine, CPUs this process may run on (affinity mask, cgroup
    quota) -- the CPU baseline states them (north_star: "timed on the same box's host cores (core count stated)")"""
    model, phys = None, set()
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model is None:
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                                        # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "physical_cores": len(phys) or None,
            "usable_cpus": usable, "cgroup_cpu_quota": quota}


class CpuSide:
    """The oracle (a port of the reference's OpenMP/popcnt path) on this box's host cores, over the SAME genome and
    the SAME index (the six sorted lists downloaded from the GPU).  Checker and reported baseline only."""

    def __init__(self, m, sym_host, frag, opts, threads):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ora
        self.ora = ora
        t0 = time.time()
        self.og = ora.Genome(sym_host, frag)
        signs, poss = [], []
        for k in range(6):
            sg, ps = m.index_export(k)          # the sorted lists in the reference's {sign, pos} form
            signs.append(sg); poss.append(ps)
        self.ix = ora.CompactIndex(self.og, opts.seedl, signs, poss)
        self.threads = threads
        self.setup_s = time.time() - t0

    def params(self, opts):
        return self.ora.make_params(seedl=opts.seedl, seedkmax=opts.seedkmax, totalkmax=opts.totalkmax, scores=opts.scores,
                                    filter_level=opts.filter_level, threads=self.threads)

    def baseline_unique(self, torch, opts, bases, qual, patl, n_reads, target_s, second_threads=8, second_s=8.0):
        """timed on a bounded strided sample of the step's reads, sized by a pilot run for about target_s of CPU work on
        self.threads threads (all the CPUs this process may use, unless --cpu-threads says otherwise); a second, shorter leg
        on `second_threads` threads (the reference's container figures are 8-thread figures) over a prefix of
        the same sample."""
        ora, p = self.ora, self.params(opts)
        k0 = min(n_reads, 50_000)
        b, q, off, _, _ = strided_sample(torch, bases, qual, n_reads, patl, k0)
        t = time.time(); ora.match_unique(self.og, self.ix, p, b, q, off); pilot = time.time() - t
        k1 = int(min(n_reads, max(k0, k0 * target_s / max(pilot, 1e-3))))
        b, q, off, stride, k1 = strided_sample(torch, bases, qual, n_reads, patl, k1)
        t = time.time(); oinfo, oscore, _ = ora.match_unique(self.og, self.ix, p, b, q, off); dt = time.time() - t
        rep = {"value": k1 / dt, "unit": "reads/s", "cores": self.threads, "kind": "port",
               "sample": "%d of the step's %d reads (strided), same %.0f Mbp genome and index (six sorted lists downloaded "
                         "from the GPU), oracle/real_oracle.c with OpenMP, %.1f s of CPU work (+%.0f s index transfer/setup)"
                         % (k1, n_reads, self.og.n / 1e6, dt, self.setup_s)}
        rep.update(host_cpu_info())
        legs = [{"threads": self.threads, "reads_per_s": k1 / dt, "reads": k1, "seconds": dt}]
        if second_threads and second_threads != self.threads and second_s > 0:
            k2 = int(min(k1, max(10_000, (k1 / dt) * second_s * second_threads / max(self.threads, 1))))
            p2 = self.ora.make_params(seedl=opts.seedl, seedkmax=opts.seedkmax, totalkmax=opts.totalkmax, scores=opts.scores,
                                      filter_level=opts.filter_level, threads=second_threads)
            t = time.time(); i2, s2, _ = ora.match_unique(self.og, self.ix, p2, b[:k2 * patl], q[:k2 * patl], off[:k2 + 1]); dt2 = time.time() - t
            legs.append({"threads": second_threads, "reads_per_s": k2 / dt2, "reads": k2, "seconds": dt2,
                         "same_records_as_first_leg": bool(np.array_equal(i2, oinfo[:k2]) and np.array_equal(s2.view(np.uint32), oscore[:k2].view(np.uint32)))})
        rep["legs"] = legs
        return rep, (stride, k1, oinfo, oscore)


# ---------------------------------------------------------------------------------------------
# timed loops
# ---------------------------------------------------------------------------------------------
def timed_unique(torch, dist, m, rlib, bases, qual, patl, n, steps, warmup, world, rank, dev, gather_dev, log=None, packed=False, fresh=False):
    """K steps of matchUnique over the resident batch, bracketed by barrier + synchronize on both sides, MAX over
    ranks.  N > 1: the records of step k travel to the root while step k+1 is matched (two alternating record
    buffers).  Returns (seconds, counters, (match_ms, launches), (repeat_ms, launches), last (info, score))."""
    info = torch.zeros(n, dtype=torch.int64, device=dev)
    score = torch.empty(n, dtype=torch.float32, device=dev)
    rg = None
    bufs = [(info, score)]
    if world > 1:
        from real_amd.distributed import RecordGatherer
        rg = RecordGatherer(n, gather_dev, scores=True, dst=0)
        bufs.append((torch.zeros_like(info), torch.empty_like(score)))
    stepno = [0]
    last = [0]
    exposed = [0.0]     # seconds this rank stood waiting for a gather inside the timed region (N > 1)

    def step():
        slot = stepno[0] % len(bufs)
        stepno[0] += 1
        last[0] = slot
        bi, bs = bufs[slot]
        if rg is not None:
            tw = time.perf_counter()
            rg.wait(slot)                                       # the gather that last read this buffer
            if gather_dev != "cpu":
                torch.cuda.current_stream().synchronize()       # (an RCCL wait only holds the stream: make the host see it)
            exposed[0] += time.perf_counter() - tw
        if not fresh:
            bi.zero_(); bs.fill_(NO_SCORE)                      # uniqueinfo(numpat): state NoMatch, score -FLT_MAX
        # (fresh: real_hip_batch.fresh -- the matcher starts every record itself, the arrays are outputs only)
        m.match_unique(bases, qual, patl=patl, info=bi, score=bs, n_reads=n, packed=packed, fresh=fresh)
        if rg is not None:                                      # the one collective: records to the root
            if gather_dev == "cpu":
                rg.start(slot, bi.cpu(), bs.cpu())
            else:
                rg.start(slot, bi, bs)

    def drain():
        if rg is not None:
            rg.wait_all()

    for _ in range(warmup):
        step()
    drain()
    if log:
        log("warmup done")
    m.counters(reset=True)
    for k in (rlib.K_MATCH_UNIQUE, rlib.K_MATCH_REPEAT):
        m.kernel_time(k, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    exposed[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    td = time.perf_counter()
    drain()                                                    # every step's records have reached the root
    if world > 1 and gather_dev != "cpu":
        torch.cuda.current_stream().synchronize()
    t_drain = time.perf_counter() - td
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gather_diag = None
    if world > 1:
        # MAX over ranks of the wall time, of the time a rank stood waiting for a gather between steps, and of the final drain
        tt = torch.tensor([dt, exposed[0], t_drain], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, w_max, d_max = (float(x) for x in tt.tolist())
        rec_bytes = 12 * n                                      # u64 record + float score per read
        gather_diag = {
            "gather_exposed_ms_per_step": (w_max + d_max) / steps * 1e3,
            "gather_wait_between_steps_ms_per_step": w_max / steps * 1e3,
            "gather_final_drain_ms": d_max * 1e3,
            "gather_bytes_per_rank_per_step": rec_bytes,
            "gather_GBps_per_link_needed": rec_bytes / (dt / steps) / 1e9,
            "gather_GBps_into_root_needed": (world - 1) * rec_bytes / (dt / steps) / 1e9,
            "note": "records of step k travel while step k+1 is matched (two alternating buffers); exposed = MAX over ranks of the time a rank "
                    "waited for a gather before reusing its buffer, plus the drain of the last step's gather, per timed step; the GB/s are what "
                    "the step rate asks of one xGMI link (peer -> root) and of the root's links together",
        }
    timed_unique.gather_diag = gather_diag
    return dt, m.counters(), m.kernel_time(rlib.K_MATCH_UNIQUE), m.kernel_time(rlib.K_MATCH_REPEAT), bufs[last[0]]


def timed_all(torch, dist, m, rlib, bases, qual, patl, n, steps, warmup, world, rank, dev, gather_dev, packed=False):
    """K steps of matchAll into device buffers; N > 1: the variable-length hit lists are gathered to the root every
    step (counts first, then payload).  Returns (seconds, counters, kernel times, hits per step on this rank,
    hits on the root per step)."""
    import ctypes as C
    from real_amd.distributed import gather_hits
    cap = 4 * n
    hits_dev = torch.empty(cap * 4, dtype=torch.int32, device=dev)
    hoff_dev = torch.empty(n + 1, dtype=torch.int64, device=dev)
    nh = [0]
    nroot = [0]

    def step():
        b = m._batch(bases, qual, None, patl, n)
        b.packed = int(bool(packed))
        nout = C.c_uint64(0)
        m.sync_inputs(bases)
        rc = m._L.real_hip_match_all(m._h, C.byref(b), hits_dev.data_ptr(), cap, C.byref(nout), hoff_dev.data_ptr())
        m._check(rc)
        nh[0] = int(nout.value)
        if world > 1:
            h = hits_dev.view(-1, 4)[:nh[0]]
            if gather_dev == "cpu":
                gh, go = gather_hits(h.cpu(), hoff_dev.cpu(), dst=0)
            else:
                gh, go = gather_hits(h, hoff_dev, dst=0)
            if rank == 0:
                nroot[0] = int(gh.shape[0])
                assert int(go[-1].item()) == nroot[0] and go.shape[0] == world * n + 1
        else:
            nroot[0] = nh[0]

    for _ in range(warmup):
        step()
    m.counters(reset=True)
    for k in (rlib.K_MATCH_ALL, rlib.K_ALL_SORT, rlib.K_MATCH_REPEAT):
        m.kernel_time(k, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kt = {"match": m.kernel_time(rlib.K_MATCH_ALL), "order": m.kernel_time(rlib.K_ALL_SORT), "repeat": m.kernel_time(rlib.K_MATCH_REPEAT)}
    return dt, m.counters(), kt, nh[0], nroot[0], (hits_dev, hoff_dev)


def roofline_block(ctr, kernel_ms, launches, patl, seedl, scores, kernel_name, hit_bytes_out=0, traffic=None):
    a_total = algorithmic_bytes(ctr, patl, seedl, scores, hit_bytes_out)
    a_launch = a_total / max(launches, 1)
    avg_ms = kernel_ms / max(launches, 1)
    achieved = a_launch / max(avg_ms * 1e-3, 1e-12) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "kernel": kernel_name, "avg_launch_ms": avg_ms, "launches": launches,
            "algorithmic_bytes_per_read": a_total / max(ctr["reads"], 1),
            "work_per_read": {k: ctr[k] / max(ctr["reads"], 1) for k in ("lookups", "probes", "candidates", "seedpass", "hits", "verified")}}


def traffic_key(mode, patl, seedl, totalk, genome_mbp, n, fmt):
    return "match_%s_%dbp_l%d_k%d_%dMbp_%dreads_%s" % (mode, patl, seedl, totalk, int(genome_mbp), n, fmt)


def recorded_traffic(mode, patl, seedl, totalk, scores, genome_mbp, n, fmt, shuffled=False):
    """HBM bytes per launch of the match kernel from the rocprofv3 PMC passes (profiles/traffic.json) -- only for a
    profiled configuration (C2, C3, C5) and only if it was measured on the kernel source that is running now."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile) or not scores or shuffled:
        return None, "not a profiled configuration"
    try:
        tj = json.load(open(tfile)).get(traffic_key(mode, patl, seedl, totalk, genome_mbp, n, fmt), {})
    except Exception:
        return None, "profiles/traffic.json unreadable"
    if not tj:
        return None, "not a profiled configuration"
    if tj.get("kernel_source_sha") != kernel_source_hash():
        return None, "profiles/traffic.json was measured on another kernel source (sha %s, running %s)" % (tj.get("kernel_source_sha"), kernel_source_hash())
    return tj.get("hbm_bytes_per_launch"), "rocprofv3 PMC passes of round %s on this kernel source: 2 x FETCH_SIZE + WRITE_SIZE" % tj.get("round")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=3000.0, help="synthetic genome size (BASELINE: 3000)")
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per GPU per step (BASELINE: 50M)")
    ap.add_argument("--patl", type=int, default=100)
    ap.add_argument("--seedl", type=int, default=32)
    ap.add_argument("--totalk", type=int, default=3)
    ap.add_argument("--scores", type=int, default=1)
    ap.add_argument("--prefix-bits", type=int, default=0)
    ap.add_argument("--table-kind", type=int, default=0, help="device bucket tables: 0 auto, 1 starts, 2 directory, 3 bucket rows (real_hip.h)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline's main leg (0 = all the CPUs this process may run on); a second leg runs on 8")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work the baseline sample is sized for (a bounded sample; 40 covers all 50M reads on 16 threads)")
    ap.add_argument("--mode", choices=["unique", "all", "ingest"], default="unique",
                    help="unique = BASELINE configs[1] (default, what the driver runs); all = configs[2] (matchAll)")
    ap.add_argument("--extras", choices=["auto", "on", "off"], default="auto",
                    help="C3 / C5 / shuffled reads / host-inclusive side measurements in the line's \"extra\" (auto: at N=1 on the BASELINE workload)")
    ap.add_argument("--extra-steps", type=int, default=5)
    ap.add_argument("--shuffle-reads", action="store_true", help="reads in random order instead of genpat's sorted order")
    ap.add_argument("--input-format", choices=["packed", "bytes"], default="packed",
                    help="the resident batch the timed region starts from: 2-bit packed bases (SURVEY 8d: pre-packed read batches; what B_io of the "
                         "algorithmic-bytes formula counts) or one mapped symbol per byte (the decoded pattern block of the reference); qualities are a byte per base in both")
    ap.add_argument("--host-buffers", action="store_true",
                    help="hand the batch over as host buffers (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank uses cuda:0 and the gather runs over gloo on host copies")
    ap.add_argument("--launch-check", action="store_true", help="only check the launcher: ranks meet over gloo on the CPU, no GPU call")
    args = ap.parse_args()

    # ---- one process per GPU: start the ranks if nobody has
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        sys.exit("bench.py: --gpus %d but the launch environment has WORLD_SIZE=%d" % (args.gpus, world))
    if args.launch_check:
        launch_check(args)
        return

    import torch
    import torch.distributed as dist
    from real_amd import lib as rlib
    from real_amd.matcher import HipMatcher, RealOptions

    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if args.rehearse_on_one_gpu else "nccl"       # nccl = RCCL over xGMI
        dist.init_process_group(backend)
    gather_dev = "cpu" if args.rehearse_on_one_gpu else dev
    world_seen = 1
    if world > 1:                                                       # the ranks count each other through the backend
        t = torch.ones(1, dtype=torch.int64, device=gather_dev)
        dist.all_reduce(t)
        world_seen = int(t.item())
        assert world_seen == dist.get_world_size() == world
    rccl_version = None
    if backend == "nccl":
        try:
            rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:      # (a label only: never worth the run)
            rccl_version = "unknown (%r)" % (e,)
    dist_info = {"world_size": world_seen, "backend": ("rccl" if backend == "nccl" else backend), "rccl_version": rccl_version}

    t_start = time.time()

    def log(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.time() - t_start, msg), file=sys.stderr, flush=True)

    G = int(args.genome_mbp * 1e6)
    opts = RealOptions(seedl=args.seedl, seedkmax=2, totalkmax=args.totalk, scores=bool(args.scores), filter_level=2).normalise()
    t_setup = time.time()
    sym = gen_genome(torch, G, 3, dev)                       # i.i.d. uniform ACGT, one fragment, seed 3
    frag = np.array([0, G], dtype=np.uint64)
    m = HipMatcher(opts, device=local, prefix_bits=args.prefix_bits, table_kind=args.table_kind)
    torch.cuda.synchronize()
    log("genome generated")
    m.set_text_symbols(0, sym, frag)
    log("text packed")
    t0 = time.time()
    n_entries, _ = m.build_index_block()
    t_index = time.time() - t0
    ibs = m.index_build_stats()
    index_build = {"wall_s": t_index, "kernel_s": ibs["kernel_ms"] / 1e3, "hipMalloc_s": ibs["alloc_ms"] / 1e3, "hipFree_s": ibs["free_ms"] / 1e3,
                   "allocated_GB": ibs["alloc_bytes"] / 1e9, "hipMalloc_calls": ibs["alloc_calls"]}
    log("index built: %d entries, prefix_bits %d, %.1f s wall = %.1f s kernels + %.1f s hipMalloc (%.0f GB) + %.1f s hipFree + rest"
        % (n_entries, m.prefix_bits, t_index, index_build["kernel_s"], index_build["hipMalloc_s"], index_build["allocated_GB"], index_build["hipFree_s"]))
    n = args.reads
    bases, qual, true_pos, true_inv = gen_reads(torch, sym, n, args.patl, 0.02, 4 + rank, dev, shuffle=args.shuffle_reads)
    log("reads generated")
    extras_on = (args.extras == "on") or (args.extras == "auto" and world == 1 and args.mode == "unique" and not args.host_buffers and
                                          (args.patl, args.seedl, args.totalk, args.scores) == (100, 32, 3, 1) and not args.shuffle_reads)
    # the CPU port keeps the six lists in host memory ((4 or 8) + 4 bytes per window and list): 144 GB for 3 Gbp with
    # 32-bit signatures, 216 GB with 64-bit ones -- the latter does not fit the box's 270 GiB with everything else
    host_index_gb = n_entries * 6 * ((4 if args.seedl <= 32 else 8) + 4) / 1e9
    if host_index_gb > 180 and not args.no_cpu_baseline:
        log("cpu baseline skipped: the CPU port's index would take %.0f GB of host memory" % host_index_gb)
        args.no_cpu_baseline = True
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    sym_host = sym.cpu().numpy() if (want_cpu or (extras_on and rank == 0)) else None
    if not extras_on:
        del sym
        torch.cuda.empty_cache()
    t_setup = time.time() - t_setup

    if args.mode == "ingest":
        ingest_side(torch, m, rlib, args, bases, qual, n, dev)
        return
    if args.host_buffers:
        host_buffers_side(torch, m, rlib, args, bases, qual, n, G)
        return

    if args.mode == "all":
        packed = args.input_format == "packed" and (n * args.patl) % 4 == 0
        dt, ctr, kt, nh, nroot, _ = timed_all(torch, dist, m, rlib, pack_bases(torch, bases, n, args.patl) if packed else bases, qual, args.patl, n,
                                              args.steps, args.warmup, world, rank, dev, gather_dev, packed=packed)
        if rank == 0:
            K = args.steps
            mk, ml = kt["match"]
            out = {"metric": "reads/sec, matchAll (all hits), 100 bp, k<=%d" % args.totalk, "value": world * n * K / dt, "unit": "reads/s",
                   "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
                   "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "side_measurement": True,
                   "config": {"workload": "matchAll, %dM synthetic %d bp FASTQ reads per GPU vs %.0f Mbp synthetic genome, k=%d, scores %s, %dxMI355X"
                                          % (n // 1_000_000, args.patl, args.genome_mbp, args.totalk, "on" if args.scores else "off", world),
                              "genome_bp": G, "reads_per_gpu_per_step": n, "read_len": args.patl,
                              "input_format": "2-bit packed bases" if packed else "one symbol per byte",
                              "hits_per_step_rank0": nh, "hits_gathered_on_root_per_step": nroot, "distributed": dist_info,
                              "parallelism": "reads sharded x%d, index replicated, hit lists gathered to rank 0 (counts first, then payload)" % world},
                   "roofline": roofline_block(ctr, mk, ml, args.patl, args.seedl, bool(args.scores),
                                              "match_kernel<W=%d,scores=%d,all,tables=%s>" % ((args.patl + 31) // 32, args.scores, TABLE_KINDS[m.table_kind]),
                                              hit_bytes_out=16 * nh * max(ml, 1),
                                              traffic=recorded_traffic("all", args.patl, args.seedl, args.totalk, args.scores, args.genome_mbp, n,
                                                                       "packed" if packed else "bytes", args.shuffle_reads)[0]),
                   "order_pass_avg_ms": kt["order"][0] / max(kt["order"][1], 1), "cpu_baseline": None}
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- the headline: C2 (or C4 = N x C2)
    packed = args.input_format == "packed" and (n * args.patl) % 4 == 0
    pk = pack_bases(torch, bases, n, args.patl) if packed else None
    dt, ctr, (match_ms, match_n), (rep_ms, rep_n), (info, score) = timed_unique(
        torch, dist, m, rlib, pk if packed else bases, qual, args.patl, n, args.steps, args.warmup, world, rank, dev, gather_dev, log, packed=packed)
    log("timed steps done: %.1f ms/step" % (dt / args.steps * 1e3))

    if rank == 0:
        K = args.steps
        value = world * n * K / dt
        st = (info >> 61) & 7
        aligned = int(((st == 1) | (st == 2)).sum().item())
        traffic, traffic_note = recorded_traffic("unique", args.patl, args.seedl, args.totalk, args.scores, args.genome_mbp, n,
                                                 "packed" if packed else "bytes", args.shuffle_reads)
        kname = "match_kernel<W=%d,scores=%d,unique,tables=%s>" % ((args.patl + 31) // 32, args.scores, TABLE_KINDS[m.table_kind])
        roof = roofline_block(ctr, match_ms, match_n, args.patl, args.seedl, bool(args.scores), kname, traffic=traffic)
        roof["traffic_source"] = traffic_note
        roof["repeat_pass_avg_ms"] = rep_ms / max(rep_n, 1)
        if world > 1 and timed_unique.gather_diag:
            dist_info.update(timed_unique.gather_diag)
        out = {
            "metric": METRIC,
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "aligned_reads_per_s": value * aligned / n,
            "value_per_gpu": value / world,
            "config": {"workload": "matchUnique, %dM synthetic %d bp FASTQ reads per GPU vs %.0f Mbp synthetic genome, "
                                   "k=%d (seed k<=2), scores %s, %dxMI355X" % (n // 1_000_000, args.patl, args.genome_mbp,
                                                                               args.totalk, "on" if args.scores else "off", world),
                       "genome_bp": G, "reads_per_gpu_per_step": n, "read_len": args.patl, "seedl": args.seedl,
                       "seedkmax": 2, "totalkmax": args.totalk, "scores": bool(args.scores), "errprob": 0.02,
                       "read_order": "shuffled" if args.shuffle_reads else "sorted by position (genpat.cpp:99)",
                       "input_format": ("2-bit packed bases (%g B per %d bp read) + one quality byte per base, resident in HBM" % (args.patl / 4, args.patl) if packed
                                        else "one mapped symbol per byte + one quality byte per base, resident in HBM"),
                       "index_entries": n_entries, "prefix_bits": m.prefix_bits, "bucket_tables": TABLE_KINDS[m.table_kind],
                       "parallelism": "reads sharded x%d, index replicated, one RCCL gather of records" % world,
                       "distributed": dist_info,
                       "uniquely_aligned_frac_rank0": aligned / n, "value_counts": "every read of the batch (aligned_reads_per_s = the uniquely aligned ones)",
                       "index_build_s": t_index, "index_build": index_build, "setup_s": t_setup,
                       "kernel_source_sha": kernel_source_hash()},
            "roofline": roof,
        }
        cpu = None
        if want_cpu:
            hc = host_cpu_info()
            all_threads = max(1, min(hc["usable_cpus"], int(hc["cgroup_cpu_quota"]) if hc["cgroup_cpu_quota"] and hc["cgroup_cpu_quota"] >= 1 else hc["usable_cpus"]))
            cpu = CpuSide(m, sym_host, frag, opts, args.cpu_threads or all_threads)
            cb, (stride, k1, oinfo, oscore) = cpu.baseline_unique(torch, opts, bases, qual, args.patl, n, args.cpu_seconds)
            # the sample doubles as a full-size parity check: GPU records of the sampled reads == CPU port
            gi = info[::stride][:k1].contiguous().cpu().numpy().view(np.uint64)
            gs = score[::stride][:k1].contiguous().cpu().numpy()
            cb["parity_on_sample"] = bool(np.array_equal(gi, oinfo) and np.array_equal(gs.view(np.uint32), oscore.view(np.uint32)))
            out["cpu_baseline"] = cb
            log("cpu baseline done: %.2f M reads/s, parity %s" % (cb["value"] / 1e6, cb["parity_on_sample"]))
        else:
            out["cpu_baseline"] = None
        if extras_on:
            del info, score
            reads = [bases, qual, true_pos, true_inv, pk]
            del bases, qual, true_pos, true_inv, pk             # (C5 needs the room: extras() frees them before its index build)
            out["extra"] = extras(torch, dist, rlib, args, m, opts, cpu, sym, sym_host, frag, reads, n, G, dev, log)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
# side measurements carried in the line's "extra" (N = 1)
# ---------------------------------------------------------------------------------------------
def extras(torch, dist, rlib, args, m, opts, cpu, sym, sym_host, frag, reads, n, G, dev, log):
    bases, qual, pk = reads[0], reads[1], reads[4]
    ex = {}
    K = max(1, args.extra_steps)
    patl = args.patl

    # (0) the same step from the other resident input format
    try:
        other_packed = pk is None and (n * patl) % 4 == 0
        ob = pack_bases(torch, bases, n, patl) if other_packed else bases
        dt, ctr, (ms, ln), (rms, rn), _ = timed_unique(torch, dist, m, rlib, ob, qual, patl, n, K, 1, 1, 0, dev, dev, packed=other_packed)
        ex["c2_packed_bases" if other_packed else "c2_byte_bases"] = {
            "ms_per_step": dt / K * 1e3, "reads_per_s": n * K / dt, "avg_match_kernel_ms": ms / max(ln, 1),
            "note": ("the headline's step with the bases as 2 bits per base" if other_packed else
                     "the headline's step with the bases as one mapped symbol per byte (the decoded pattern block of the reference; the round-1 headline format)")}
        del ob
        log("extra: %s bases %.1f ms/step" % ("packed" if other_packed else "byte", dt / K * 1e3))
    except Exception as e:
        ex["c2_other_input_format"] = {"error": repr(e)}

    # (0b) the headline's step with the records started by the matcher itself (real_hip_batch.fresh, what the `real` driver does for
    # the first genome block): no initialisation pass over the record arrays, and the kernel does not read them
    try:
        hb = pk if pk is not None else bases
        dt, ctr, (ms, ln), (rms, rn), (fi, fs) = timed_unique(torch, dist, m, rlib, hb, qual, patl, n, K, 1, 1, 0, dev, dev, packed=pk is not None, fresh=True)
        ref_i = torch.zeros(n, dtype=torch.int64, device=dev)
        ref_s = torch.full((n,), NO_SCORE, dtype=torch.float32, device=dev)
        m.match_unique(hb, qual, patl=patl, info=ref_i, score=ref_s, n_reads=n, packed=pk is not None)
        ex["c2_fresh_records"] = {"ms_per_step": dt / K * 1e3, "reads_per_s": n * K / dt, "avg_match_kernel_ms": ms / max(ln, 1),
                                  "records_equal_initialised_run": bool(torch.equal(fi, ref_i) and torch.equal(fs.view(torch.int32), ref_s.view(torch.int32))),
                                  "note": "real_hip_batch.fresh = 1: the record arrays are outputs only (the headline initialises them every step and the kernel reads them)"}
        del fi, fs, ref_i, ref_s
        log("extra: fresh records %.1f ms/step" % (dt / K * 1e3))
    except Exception as e:
        ex["c2_fresh_records"] = {"error": repr(e)}

    # (1) the same step on the same reads in shuffled order (a sequencer does not sort; genpat does)
    try:
        perm = torch.randperm(n, device=dev)
        sb = bases.view(n, patl)[perm].contiguous().view(-1)
        sq = qual.view(n, patl)[perm].contiguous().view(-1)
        del perm
        spk = pk is not None
        if spk:
            sb = pack_bases(torch, sb, n, patl)
        dt, ctr, (ms, ln), (rms, rn), _ = timed_unique(torch, dist, m, rlib, sb, sq, patl, n, K, 1, 1, 0, dev, dev, packed=spk)
        ex["c2_shuffled_reads"] = {"ms_per_step": dt / K * 1e3, "reads_per_s": n * K / dt, "avg_match_kernel_ms": ms / max(ln, 1),
                                   "note": "the step's 50M reads in a random order: consecutive lanes verify unrelated text"}
        del sb, sq
        torch.cuda.empty_cache()
        log("extra: shuffled reads %.1f ms/step" % (dt / K * 1e3))
    except Exception as e:      # a side measurement must not take the headline down
        ex["c2_shuffled_reads"] = {"error": repr(e)}

    # (2) host-inclusive C2: pinned, double-buffered, packed batches through real_hip_match_unique_submit
    try:
        ex["c2_host_inclusive"] = host_inclusive(torch, m, rlib, bases, qual, patl, n, log)
    except Exception as e:
        ex["c2_host_inclusive"] = {"error": repr(e)}

    # (3) C3: matchAll, k=2, the same index and reads
    try:
        m.set_match_params(totalkmax=2)
        dt, ctr, kt, nh, _, (hits_dev, hoff_dev) = timed_all(torch, dist, m, rlib, pk if pk is not None else bases, qual, patl, n, K, 1, 1, 0, dev, dev,
                                                             packed=pk is not None)
        mk, ml = kt["match"]
        c3 = {"workload": "matchAll, %dM synthetic %d bp reads vs %.0f Mbp genome, k=2, scores on (BASELINE configs[2])" % (n // 1_000_000, patl, args.genome_mbp),
              "ms_per_step": dt / K * 1e3, "reads_per_s": n * K / dt, "hits_per_step": nh,
              "input_format": "2-bit packed bases" if pk is not None else "one symbol per byte",
              "order_pass_avg_ms": kt["order"][0] / max(kt["order"][1], 1),
              "roofline": roofline_block(ctr, mk, ml, patl, args.seedl, True,
                                         "match_kernel<W=%d,scores=1,all,tables=%s>" % ((patl + 31) // 32, TABLE_KINDS[m.table_kind]),
                                         hit_bytes_out=16 * nh * max(ml, 1),
                                         traffic=recorded_traffic("all", patl, args.seedl, 2, 1, args.genome_mbp, n, "packed" if pk is not None else "bytes")[0])}
        if cpu is not None:     # parity on a sample: the oracle's unified hit lists of 200k of the step's reads == the device's
            b, q, off, stride, k1 = strided_sample(torch, bases, qual, n, patl, 200_000)
            p2 = cpu.params(m.opts)
            oh, ooff, _ = cpu.ora.match_all(cpu.og, cpu.ix, p2, b, q, off)
            ho = hoff_dev[::stride][:k1].cpu().numpy()
            hn = (hoff_dev[1:][::stride][:k1] - hoff_dev[:-1][::stride][:k1]).cpu().numpy()
            same = np.array_equal(hn.astype(np.int64), np.diff(ooff.astype(np.int64)))
            if same and oh.shape[0]:
                idx = torch.from_numpy(np.concatenate([np.arange(a, a + c) for a, c in zip(ho.tolist(), hn.tolist())]).astype(np.int64)).to(dev)
                gh = hits_dev.view(-1, 4)[idx].cpu().numpy().reshape(-1).view(rlib.HIT_DTYPE)
                same = (np.array_equal(gh["pos"], oh["pos"]) and np.array_equal(gh["k"], oh["k"]) and np.array_equal(gh["inverted"], oh["inverted"]) and
                        np.array_equal(gh["frag"], oh["frag"].astype(np.uint16)) and np.array_equal(gh["score"].view(np.uint32), oh["score"].view(np.uint32)))
            c3["parity_on_sample"] = bool(same)
            c3["parity_sample"] = "%d reads (strided), %d hits: per-read hit counts, order, pos, strand, k, fragment and score bits vs the oracle" % (k1, oh.shape[0])
        ex["c3_match_all"] = c3
        del hits_dev, hoff_dev
        log("extra: C3 %.1f ms/step, %d hits" % (dt / K * 1e3, nh))
    except Exception as e:
        ex["c3_match_all"] = {"error": repr(e)}
    finally:
        m.set_match_params(totalkmax=opts.totalkmax)

    # (4) C5: 150 bp, 64-bit signatures, k=5, scores on -- the index is rebuilt
    try:
        from real_amd.matcher import HipMatcher, RealOptions
        if cpu is not None:
            cpu.ix = None           # 144 GB of host lists
        m.close()
        del bases, qual
        reads.clear()
        torch.cuda.empty_cache()
        o5 = RealOptions(seedl=64, seedkmax=2, totalkmax=5, scores=True, filter_level=2).normalise()
        m5 = HipMatcher(o5, device=dev.index)
        m5.set_text_symbols(0, sym, frag)
        t0 = time.time()
        ne5, _ = m5.build_index_block()
        t_ix5 = time.time() - t0
        ibs5 = m5.index_build_stats()
        b5, q5, pos5, inv5 = gen_reads(torch, sym, n, 150, 0.02, 12, dev)
        p5 = pk is not None and (n * 150) % 4 == 0
        dt, ctr, (ms, ln), (rms, rn), (info5, score5) = timed_unique(torch, dist, m5, rlib, pack_bases(torch, b5, n, 150) if p5 else b5, q5, 150, n, K, 1, 1, 0,
                                                                     dev, dev, packed=p5)
        st = (info5 >> 61) & 7
        al = ((st == 1) | (st == 2))
        c5 = {"workload": "matchUnique, %dM synthetic 150 bp reads vs %.0f Mbp genome, seedl 64 (64-bit signatures), k=5, scores on (BASELINE configs[4])" % (n // 1_000_000, args.genome_mbp),
              "ms_per_step": dt / K * 1e3, "reads_per_s": n * K / dt, "uniquely_aligned_frac": float(al.float().mean().item()),
              "input_format": "2-bit packed bases" if p5 else "one symbol per byte",
              "index_build_s": t_ix5, "index_build_kernel_s": ibs5["kernel_ms"] / 1e3, "index_build_hipMalloc_s": ibs5["alloc_ms"] / 1e3,
              "index_build_hipFree_s": ibs5["free_ms"] / 1e3, "bucket_tables": TABLE_KINDS[m5.table_kind], "prefix_bits": m5.prefix_bits,
              "roofline": roofline_block(ctr, ms, ln, 150, 64, True, "match_kernel<W=5,scores=1,unique,tables=%s>" % TABLE_KINDS[m5.table_kind],
                                         traffic=recorded_traffic("unique", 150, 64, 5, 1, args.genome_mbp, n, "packed" if p5 else "bytes")[0]),
              "repeat_pass_avg_ms": rms / max(rn, 1)}
        # check on a sample against the oracle's text functions (its 64-bit index does not fit the host beside everything else):
        # every uniquely aligned read of the sample sits where it was cut from, with the mismatch count and the score bits the
        # oracle computes for that position and strand
        if sym_host is not None:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import ctypes as C
            import oracle_lib as ora
            og = cpu.og if cpu is not None else ora.Genome(sym_host, frag)
            b, q, off, stride, k1 = strided_sample(torch, b5, q5, n, 150, 100_000)
            gi = info5[::stride][:k1].cpu().numpy().view(np.uint64)
            gs = score5[::stride][:k1].cpu().numpy()
            tp = pos5[::stride][:k1].cpu().numpy()
            ti = inv5[::stride][:k1].cpu().numpy()
            state, _, err, _, pos = ora.unpack_record(gi)
            LL, _ = ora.scoring_table()
            bad = 0
            nal = 0
            for i in np.nonzero((state == 1) | (state == 2))[0].tolist():
                nal += 1
                rb, rq = b[150 * i:150 * (i + 1)], q[150 * i:150 * (i + 1)]
                inv = int(state[i] == 2)
                ob = (3 - rb[::-1]) if inv else rb
                k = int((ob != sym_host[pos[i]:pos[i] + 150]).sum())
                sc = np.float32(ora.lib().ora_compute_score(og.h, LL.ctypes.data, inv, rb.ctypes.data, rq.ctypes.data, int(pos[i]), 150))
                if pos[i] != tp[i] or inv != ti[i] or k != err[i] or sc.view(np.uint32) != gs[i:i + 1].view(np.uint32)[0]:
                    bad += 1
            c5["verified_on_sample"] = bool(bad == 0 and nal > 0)
            c5["verify_sample"] = ("%d reads (strided), %d uniquely aligned: position and strand = where the read was cut from, mismatches and score bits "
                                   "recomputed by the oracle's text functions; %d disagree" % (k1, nal, bad))
        ex["c5_150bp_l64"] = c5
        log("extra: C5 %.1f ms/step" % (dt / K * 1e3))
        m5.close()
    except Exception as e:
        ex["c5_150bp_l64"] = {"error": repr(e)}
    return ex


def host_inclusive(torch, m, rlib, bases, qual, patl, n, log, steps=2, chunk=5_000_000):
    """C2 with the batch coming from HOST memory every step: pinned buffers, 2-bit packed bases + one quality byte
    per base (125 B/read in, 12 B/read of records out), chunks of 5 M reads through the two slots of
    real_hip_match_unique_submit -- chunk k+1 crosses PCIe while chunk k is matched."""
    assert patl % 4 == 0 and chunk % 8 == 0
    bpr = patl // 4
    hb = m.host_alloc((n * bpr,), np.uint8)
    hq = m.host_alloc((n * patl,), np.uint8)
    hi = m.host_alloc((n,), np.uint64)
    hs = m.host_alloc((n,), np.float32)
    # pack on the device (4 bases per byte, MSB first), bring both arrays to the pinned buffers once
    pk = pack_bases(torch, bases, n, patl)
    torch.from_numpy(hb).copy_(pk)
    torch.from_numpy(hq).copy_(qual)
    del pk
    torch.cuda.synchronize()
    cuts = list(range(0, n, chunk))

    def one_pass():
        for k, lo in enumerate(cuts):
            hi_ = min(n, lo + chunk)
            m.wait(k % 2)
            m.submit_unique(k % 2, hb[lo * bpr:hi_ * bpr], hq[lo * patl:hi_ * patl], hi[lo:hi_], hs[lo:hi_], patl=patl, n_reads=hi_ - lo,
                            packed=True, fresh=True)
        m.wait(0); m.wait(1)

    one_pass()                                                  # warm-up (slot buffers)
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    dt = (time.perf_counter() - t0) / steps
    # the records that came back == the resident-input run's
    ri = torch.zeros(n, dtype=torch.int64, device=bases.device)
    rs = torch.full((n,), NO_SCORE, dtype=torch.float32, device=bases.device)
    m.match_unique(bases, qual, patl=patl, info=ri, score=rs, n_reads=n)
    same = bool(np.array_equal(ri.cpu().numpy().view(np.uint64), np.asarray(hi)) and np.array_equal(rs.cpu().numpy().view(np.uint32), np.asarray(hs).view(np.uint32)))
    h2d = n * (bpr + patl) / dt / 1e9
    out = {"reads_per_s": n / dt, "ms_per_step": dt * 1e3, "chunk_reads": chunk, "slots": 2,
           "input": "pinned host memory, 2-bit packed bases + 1 quality byte per base = %d B/read; records initialised on the device, 12 B/read back" % (bpr + patl),
           "records_equal_resident_run": same,
           "pcie_roofline": {"bound": "pcie", "achieved": h2d, "peak": 64.0, "unit": "GB/s", "frac": h2d / 64.0,
                             "note": "host-to-device bytes per second against PCIe Gen5 x16 (64 GB/s per direction)"}}
    if log:
        log("extra: host-inclusive %.0f M reads/s (%.1f GB/s over PCIe), records equal: %s" % (n / dt / 1e6, h2d, same))
    return out


def ingest_side(torch, m, rlib, args, bases, qual, n, dev):
    """side measurement of read ingestion on the device: FASTQ text of the step's reads, resident
    in HBM, parsed by real_hip_parse_reads into batch arrays; then matched from those arrays"""
    nn = min(n, (4 * 2**30 - 2**20) // (2 * args.patl + 16))       # one text chunk stays under 4 GiB
    R = 2 * args.patl + 16
    rec = torch.empty((nn, R), dtype=torch.uint8, device=dev)
    idx = torch.arange(nn, device=dev, dtype=torch.int64)
    rec[:, 0] = ord("@")
    for d in range(10):
        rec[:, 1 + d] = (48 + (idx // 10 ** (9 - d)) % 10).to(torch.uint8)
    lut = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8, device=dev)
    P = args.patl
    rec[:, 11] = 10
    rec[:, 12:12 + P] = lut[bases[:nn * P].view(nn, P).long()]
    rec[:, 12 + P] = 10; rec[:, 13 + P] = ord("+"); rec[:, 14 + P] = 10
    rec[:, 15 + P:15 + 2 * P] = qual[:nn * P].view(nn, P) + 33
    rec[:, 15 + 2 * P] = 10
    text = rec.view(-1)
    del idx
    torch.cuda.synchronize()
    p = m.parse_reads(text, fastq=True, quality_offset=33)                       # warm-up (allocations)
    ok = (p.n_reads == nn and np.array_equal(m.download(p.bases, 10_000_000, np.uint8), bases[:10_000_000].cpu().numpy())
          and np.array_equal(m.download(p.qual, 10_000_000, np.uint8), qual[:10_000_000].cpu().numpy()))
    m.kernel_time(rlib.K_PARSE, reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        p = m.parse_reads(text, fastq=True, quality_offset=33)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    hinfo, hscore = m.match_unique_parsed(p)
    st = (hinfo >> np.uint64(61)) & np.uint64(7)
    print(json.dumps({"side_measurement": True, "mode": "ingest", "reads": nn, "text_bytes": int(text.numel()),
                      "parse_ms": dt * 1e3, "parse_ms_by_hip_events": m.kernel_time(rlib.K_PARSE)[0] / max(m.kernel_time(rlib.K_PARSE)[1], 1),
                      "reads_per_s": nn / dt, "text_GB_per_s": text.numel() / dt / 1e9,
                      "hbm_bytes_algorithmic": int(text.numel()) + 2 * nn * P + 8 * nn, "parsed_equals_source_on_10M_symbols": bool(ok),
                      "uniquely_aligned_frac_from_parsed": float(((st == 1) | (st == 2)).mean())}), flush=True)


def host_buffers_side(torch, m, rlib, args, bases, qual, n, G):
    """manual side measurement: the batch handed over as pageable host buffers through the synchronous entry points"""
    hb = bases.cpu().numpy()
    hq = qual.cpu().numpy()
    nh = 0

    def side_step():
        nonlocal nh
        if args.mode == "unique":
            m.match_unique(hb, hq, patl=args.patl, n_reads=n)
        else:
            h, o = m.match_all(hb, hq, patl=args.patl, n_reads=n, cap=4 * n)
            nh = h.shape[0]

    for _ in range(args.warmup):
        side_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        side_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"side_measurement": True, "mode": args.mode, "host_buffers": "pageable, unpacked, synchronous",
                      "reads_per_s": n * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "hits_per_step": nh,
                      "genome_bp": G, "reads": n, "read_len": args.patl, "seedl": args.seedl, "totalk": args.totalk}), flush=True)


if __name__ == "__main__":
    main()
