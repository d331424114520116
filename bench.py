#!/usr/bin/env python3
"""bench.py -- REAL read-matching hot path on MI355X.

One step = one pass of the hot path (pack -> signatures -> bucket lookup -> seed
filter -> verify -> score -> best/unique fold) over one batch of synthetic reads
that is already resident in HBM.  Workload = BASELINE.json configs[1]:
matchUnique, 50M synthetic 100 bp FASTQ reads vs a 3 Gbp synthetic genome,
k=3 (seed k<=2), scores on.  With --gpus N every rank holds the whole index
(replicated) and its own shard of reads (weak scaling, C4 = 8 x 50M); the only
data-path communication is one RCCL gather of the per-read records to rank 0.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(c, patl, seedl, scores):
    """A = R*B_io + L*8 + P*w + C*(8+w) + S*(3*18+8+8*W_rest) + H*(18+8*W_score),
    evaluated with the kernel's own work counters (P = index entries examined)."""
    w = 4 if seedl <= 32 else 8
    w_rest = -(-(patl - seedl) // 32) + 1
    w_score = (-(-patl // 32) + 1) if scores else 0
    b_io = -(-patl // 4) + (patl if scores else 0) + 2 * (8 + (4 if scores else 0))
    return (c["reads"] * b_io + c["lookups"] * 8 + c["probes"] * w + c["candidates"] * (8 + w) +
            c["seedpass"] * (3 * 18 + 8 + 8 * w_rest) + c["hits"] * (18 + 8 * w_score))


def _synth():
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "real_amd", "libreal_synth.so"))
    L.real_synth_genome.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.real_synth_positions.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.real_synth_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_double, C.c_uint64,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def gen_genome(torch, G, seed, device):
    """randstr.cpp:27-53: i.i.d. uniform ACGT, generated on the device (bench_support/synth_kernels.hip)."""
    sym = torch.empty(G, dtype=torch.uint8, device=device)
    torch.cuda.synchronize()
    rc = _synth().real_synth_genome(sym.data_ptr(), G, seed)
    assert rc == 0, "synth_genome failed: hip error %d" % rc
    return sym


def gen_reads(torch, sym, n_reads, patl, errprob, seed, device):
    """genpat's distribution (genpat.cpp:96-157) on the device: sorted uniform start
    positions, strand flip p=0.5, per-base substitution to a different base with
    probability errprob, quality 35 ('D'-33) unchanged / 9 ('*'-33) mutated."""
    L = _synth()
    n = sym.shape[0]
    pos = torch.empty(n_reads, dtype=torch.int64, device=device)
    torch.cuda.synchronize()
    rc = L.real_synth_positions(pos.data_ptr(), n_reads, n - patl + 1, seed)
    assert rc == 0
    pos, _ = torch.sort(pos)          # genpat sorts the sampled positions (genpat.cpp:99)
    bases = torch.empty(n_reads * patl, dtype=torch.uint8, device=device)
    qual = torch.empty(n_reads * patl, dtype=torch.uint8, device=device)
    torch.cuda.synchronize()
    rc = L.real_synth_reads(sym.data_ptr(), pos.data_ptr(), n_reads, patl, errprob, seed, bases.data_ptr(), qual.data_ptr(), None)
    assert rc == 0, "synth_reads failed: hip error %d" % rc
    return bases, qual, pos


def cpu_baseline(torch, m, sym_host, frag, opts, bases, qual, patl, n_reads, threads, target_s=15.0):
    """The oracle (a port of the reference's OpenMP/popcnt path) timed on this box's host
    cores on a bounded sample of the SAME workload: same genome, same index (the six sorted
    lists downloaded from the GPU), a strided sample of the same reads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ora
    t0 = time.time()
    og = ora.Genome(sym_host, frag)
    signs, poss = [], []
    for k in range(6):
        sg, ps = m.index_export(k)          # the sorted lists in the reference's {sign, pos} form
        signs.append(sg); poss.append(ps)
    ix = ora.CompactIndex(og, opts.seedl, signs, poss)
    p = ora.make_params(seedl=opts.seedl, seedkmax=opts.seedkmax, totalkmax=opts.totalkmax, scores=opts.scores,
                        filter_level=opts.filter_level, threads=threads)
    setup_s = time.time() - t0

    def sample(k):
        stride = max(1, n_reads // k)           # strided slice: no gather kernels on >2^31-element tensors
        b = bases.view(n_reads, patl)[::stride][:k].contiguous().cpu().numpy().reshape(-1)
        q = qual.view(n_reads, patl)[::stride][:k].contiguous().cpu().numpy().reshape(-1)
        k = b.shape[0] // patl
        off = (np.arange(k + 1, dtype=np.uint64) * np.uint64(patl))
        return b, q, off, (stride, k)

    k0 = min(n_reads, 50_000)
    b, q, off, _ = sample(k0)
    t = time.time(); ora.match_unique(og, ix, p, b, q, off); pilot = time.time() - t
    k1 = int(min(n_reads, max(k0, k0 * target_s / max(pilot, 1e-3))))
    b, q, off, idx = sample(k1)
    k1 = idx[1]
    t = time.time(); oinfo, oscore, octr = ora.match_unique(og, ix, p, b, q, off); dt = time.time() - t
    return {"value": k1 / dt, "unit": "reads/s", "cores": threads, "kind": "port",
            "sample": "%d of the step's %d reads (strided), same %.0f Mbp genome and index (six sorted lists downloaded "
                      "from the GPU), oracle/real_oracle.c with OpenMP, %.1f s of CPU work (+%.0f s index transfer/setup)"
                      % (k1, n_reads, og.n / 1e6, dt, setup_s)}, (idx, oinfo, oscore)


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
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work the baseline sample is sized for (a bounded sample; 40 covers all 50M reads on 16 threads)")
    ap.add_argument("--mode", choices=["unique", "all", "ingest"], default="unique",
                    help="unique = BASELINE configs[1] (default, what the driver runs); all = configs[2] (matchAll, manual runs)")
    ap.add_argument("--host-buffers", action="store_true",
                    help="hand the batch over as host buffers (PCIe-inclusive rate for DESIGN.md; never the headline value)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank uses cuda:0 and the gather runs over gloo on host copies")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from real_amd import lib as rlib
    from real_amd.matcher import AllMatcher, RealOptions, UniqueMatcher
    from real_amd.distributed import gather_records

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.rehearse_on_one_gpu else "nccl")     # nccl = RCCL over xGMI

    G = int(args.genome_mbp * 1e6)
    opts = RealOptions(seedl=args.seedl, seedkmax=2, totalkmax=args.totalk, scores=bool(args.scores), filter_level=2).normalise()
    def log(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.time() - t_start, msg), file=sys.stderr, flush=True)
    t_start = time.time()
    t_setup = time.time()
    sym = gen_genome(torch, G, 3, dev)                       # i.i.d. uniform ACGT, one fragment, seed 3
    frag = np.array([0, G], dtype=np.uint64)
    m = (UniqueMatcher if args.mode == "unique" else AllMatcher)(opts, device=local, prefix_bits=args.prefix_bits, table_kind=args.table_kind)
    torch.cuda.synchronize()
    log("genome generated")
    m.set_text_symbols(0, sym, frag)
    log("text packed")
    t0 = time.time()
    n_entries, _ = m.build_index_block()
    t_index = time.time() - t0
    log("index built: %d entries, prefix_bits %d, %.1f s" % (n_entries, m.prefix_bits, t_index))
    n = args.reads
    bases, qual, true_pos = gen_reads(torch, sym, n, args.patl, 0.02, 4 + rank, dev)
    log("reads generated")
    # the CPU port keeps the six lists in host memory ((4 or 8) + 4 bytes per window and list): 144 GB for 3 Gbp with
    # 32-bit signatures, 216 GB with 64-bit ones -- the latter does not fit the box's 270 GiB with everything else
    host_index_gb = n_entries * 6 * ((4 if args.seedl <= 32 else 8) + 4) / 1e9
    if host_index_gb > 180 and not args.no_cpu_baseline:
        log("cpu baseline skipped: the CPU port's index would take %.0f GB of host memory" % host_index_gb)
        args.no_cpu_baseline = True
    sym_host = sym.cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    del sym
    torch.cuda.empty_cache()
    info = torch.zeros(n, dtype=torch.int64, device=dev)
    score = torch.empty(n, dtype=torch.float32, device=dev)
    t_setup = time.time() - t_setup

    if args.mode == "ingest":
        # side measurement of read ingestion on the device: FASTQ text of the step's reads, resident
        # in HBM, parsed by real_hip_parse_reads into batch arrays; then matched from those arrays
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
        return

    if args.mode == "all" or args.host_buffers:
        # manual side measurements (matchAll; host buffers): their own simple loop and JSON line
        import ctypes as C
        from real_amd.lib import HIT_DTYPE, RealHipBatch
        hb = bases.cpu().numpy() if args.host_buffers else None
        hq = qual.cpu().numpy() if args.host_buffers else None
        cap = 4 * n
        hits_dev = torch.empty(cap * 16, dtype=torch.uint8, device=dev) if not args.host_buffers else None
        hoff_dev = torch.empty(n + 1, dtype=torch.int64, device=dev) if not args.host_buffers else None
        nh = 0

        def side_step():
            nonlocal nh
            if args.mode == "unique":
                hi, hs = m.match_unique(hb, hq, patl=args.patl, n_reads=n)
                return
            if args.host_buffers:
                h, o = m.match_all(hb, hq, patl=args.patl, n_reads=n, cap=cap)
                nh = h.shape[0]
                return
            b = m._batch(bases, qual, None, args.patl, n)
            nout = C.c_uint64(0)
            rc = m._L.real_hip_match_all(m._h, C.byref(b), hits_dev.data_ptr(), cap, C.byref(nout), hoff_dev.data_ptr())
            m._check(rc)
            nh = int(nout.value)

        for _ in range(args.warmup):
            side_step()
        for kk in range(5):
            m.kernel_time(kk, reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            side_step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"side_measurement": True, "mode": args.mode, "host_buffers": bool(args.host_buffers),
                          "reads_per_s": n * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "hits_per_step": nh,
                          "genome_bp": G, "reads": n, "read_len": args.patl, "seedl": args.seedl, "totalk": args.totalk,
                          "kernel_ms": {nm: m.kernel_time(i)[0] / max(m.kernel_time(i)[1], 1)
                                        for i, nm in enumerate(["match_unique", "match_all", "all_sort", "index", "match_repeat"])}}), flush=True)
        return

    # N > 1: the records of step k travel to the root while step k+1 is matched (two alternating record buffers)
    rg = None
    bufs = [(info, score)]
    if world > 1:
        from real_amd.distributed import RecordGatherer
        rg = RecordGatherer(n, "cpu" if args.rehearse_on_one_gpu else dev, scores=True, dst=0)
        bufs.append((torch.zeros_like(info), torch.empty_like(score)))
    stepno = [0]

    def step():
        slot = stepno[0] % len(bufs)
        stepno[0] += 1
        bi, bs = bufs[slot]
        if rg is not None:
            rg.wait(slot)                                       # the gather that last read this buffer
        bi.zero_(); bs.fill_(-3.4028234663852886e38)           # uniqueinfo(numpat): state NoMatch, score -FLT_MAX
        torch.cuda.current_stream().synchronize()
        m.match_unique(bases, qual, patl=args.patl, info=bi, score=bs, n_reads=n)
        if rg is not None:                                      # the one collective: records to the root
            if args.rehearse_on_one_gpu:
                rg.start(slot, bi.cpu(), bs.cpu())
            else:
                rg.start(slot, bi, bs)

    def drain():
        if rg is not None:
            rg.wait_all()

    for _ in range(args.warmup):
        step()
    drain()
    log("warmup done")
    m.counters(reset=True)
    for k in (rlib.K_MATCH_UNIQUE, rlib.K_MATCH_REPEAT):
        m.kernel_time(k, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                                    # every step's records have reached the root
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=("cpu" if args.rehearse_on_one_gpu else dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    log("timed steps done: %.1f ms/step" % (dt / args.steps * 1e3))
    ctr = m.counters()
    match_ms, match_n = m.kernel_time(rlib.K_MATCH_UNIQUE)
    rep_ms, rep_n = m.kernel_time(rlib.K_MATCH_REPEAT)

    if rank == 0:
        K = args.steps
        value = world * n * K / dt
        a_total = algorithmic_bytes(ctr, args.patl, args.seedl, bool(args.scores))
        a_launch = a_total / max(match_n, 1)
        avg_ms = match_ms / max(match_n, 1)
        achieved = a_launch / (avg_ms * 1e-3) / 1e9
        st = (info.view(torch.int64) >> 61) & 7
        aligned = int(((st == 1) | (st == 2)).sum().item())
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                key = "match_unique_%dMbp_%dreads" % (int(args.genome_mbp), n)
                if (args.patl, args.seedl, args.totalk, args.scores) == (100, 32, 3, 1):     # the profiled configuration only
                    traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "aligned reads/sec (100 bp, k≤3) at 1/2/4/8 MI355X; achieved HBM GB/s vs peak",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "matchUnique, %dM synthetic %d bp FASTQ reads per GPU vs %.0f Mbp synthetic genome, "
                                   "k=%d (seed k<=2), scores %s, %dxMI355X" % (n // 1_000_000, args.patl, args.genome_mbp,
                                                                               args.totalk, "on" if args.scores else "off", world),
                       "genome_bp": G, "reads_per_gpu_per_step": n, "read_len": args.patl, "seedl": args.seedl,
                       "seedkmax": 2, "totalkmax": args.totalk, "scores": bool(args.scores), "errprob": 0.02,
                       "index_entries": n_entries, "prefix_bits": m.prefix_bits, "bucket_tables": ("starts", "digest", "fingerprint", "rows")[m.table_kind],
                       "parallelism": "reads sharded x%d, index replicated, one RCCL gather of records" % world,
                       "uniquely_aligned_frac_rank0": aligned / n, "index_build_s": t_index, "setup_s": t_setup},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "match_kernel<W=%d,scores=%d,unique,tables=%s>" % ((args.patl + 31) // 32, args.scores, ("starts", "digest", "fingerprint", "rows")[m.table_kind]), "avg_launch_ms": avg_ms, "launches": match_n,
                         "algorithmic_bytes_per_read": a_total / max(ctr["reads"], 1),
                         "repeat_pass_avg_ms": rep_ms / max(rep_n, 1),
                         "work_per_read": {k: ctr[k] / max(ctr["reads"], 1) for k in ("lookups", "probes", "candidates", "seedpass", "hits", "verified")}},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, (idx, oinfo, oscore) = cpu_baseline(torch, m, sym_host, frag, opts, bases, qual, args.patl, n, args.cpu_threads, target_s=args.cpu_seconds)
            # the sample doubles as a full-size parity check: GPU records of the sampled reads == CPU port
            stride, k1 = idx
            gi = info[::stride][:k1].contiguous().cpu().numpy().view(np.uint64)
            gs = score[::stride][:k1].contiguous().cpu().numpy()
            cb["parity_on_sample"] = bool(np.array_equal(gi, oinfo) and np.array_equal(gs.view(np.uint32), oscore.view(np.uint32)))
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
