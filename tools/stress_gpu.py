#!/usr/bin/env python3
"""Developer tool: re-run the seeded GPU parity tests (fuzzed chains against the oracle) with shifted seeds.
usage: stress_gpu.py [rounds]   -- run on the GPU box"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
import libxaac_amd  # noqa: E402
import test_imdct_gpu as ti  # noqa: E402
import test_limiter_gpu as tl  # noqa: E402
import test_sbr_gpu as ts  # noqa: E402
import test_sbr_hq_gpu as th  # noqa: E402
import test_esbr_sbr_gpu as te  # noqa: E402
import test_usac_imdct as tu  # noqa: E402
import test_esbr_qmf as tq  # noqa: E402
import test_imdct960_gpu as t9  # noqa: E402
import test_imdct_ld_gpu as tld  # noqa: E402
import test_qmf_eld_gpu as teld  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
oracle = oracle_lib.load_oracle()
real_rng = np.random.default_rng
bad = 0
for r in range(1, rounds + 1):
    np.random.default_rng = lambda seed=None, _r=r: real_rng(None if seed is None else seed + 7919 * _r)
    ctx0 = libxaac_amd.XaacContext(0, 0)
    ctx = libxaac_amd.XaacContext(0)
    jobs = [
        ("imdct mixed cf1", lambda: ti.test_mixed_batch_and_interleave(ctx, oracle, 1)),
        ("imdct mixed cf2", lambda: ti.test_mixed_batch_and_interleave(ctx, oracle, 2)),
        ("imdct chain", lambda: ti.test_stream_chain_state_carried_on_device(ctx, oracle)),
        ("sbr lp fuzz chain", lambda: ts.test_fuzzed_chain_vs_oracle_stereo_interleaved(ctx0, oracle)),
        ("sbr hq/ps fuzz chain", lambda: th.test_fuzzed_chain_vs_oracle(ctx0, oracle)),
        ("sbr hq mono", lambda: th.test_hq_mono_without_ps_vs_oracle(ctx0, oracle)),
        ("limiter chains 48k", lambda: tl.test_chains_vs_oracle(ctx, oracle, 2, 48000, 1024)),
        ("limiter chains 44k1 mono", lambda: tl.test_chains_vs_oracle(ctx, oracle, 1, 44100, 1024)),
        ("limiter chains 8k", lambda: tl.test_chains_vs_oracle(ctx, oracle, 1, 8000, 1024)),
        ("limiter stale idx", lambda: tl.test_state_with_stale_max_idx(ctx, oracle)),
        ("limiter planar", lambda: tl.test_planar_block_layout(ctx, oracle, 2, 1024)),
        ("eSBR chain (Path A)", lambda: te.test_chain_vs_oracle(oracle)),
        ("eSBR + float PS chain", lambda: te.test_ps_chain_vs_oracle(oracle)),
        ("eSBR banks", lambda: tq.test_gpu_analysis_then_synthesis_vs_oracle(oracle)),
        ("USAC IMDCT batch 1024", lambda: tu.test_gpu_large_batch_vs_oracle(oracle, 1024)),
        ("USAC IMDCT batch 768", lambda: tu.test_gpu_large_batch_vs_oracle(oracle, 768)),
        ("eSBR chain with harmonic SBR", lambda: te.test_chain_with_harmonic_transposer_vs_oracle(oracle)),
        ("960-line IMDCT, every transition", lambda: t9.test_every_transition_vs_oracle(oracle, 0)),
        ("960-line IMDCT, stereo walk", lambda: t9.test_stereo_walk_with_state_on_device(oracle)),
        ("LD IMDCT 512", lambda: tld.test_stereo_chains_vs_oracle(oracle, 512, 0)),
        ("ELD IMDCT 512", lambda: tld.test_stereo_chains_vs_oracle(oracle, 512, 1)),
        ("LD IMDCT 480", lambda: tld.test_stereo_chains_vs_oracle(oracle, 480, 0)),
        ("ELD IMDCT 480", lambda: tld.test_stereo_chains_vs_oracle(oracle, 480, 1)),
        ("ELD analysis bank", lambda: teld.test_eld_analysis_chain_vs_oracle(oracle, 15)),
        ("ELD synthesis bank", lambda: teld.test_eld_synthesis_chain_vs_oracle(oracle, 16)),
    ]
    for name, job in jobs:
        try:
            job()
        except Exception:
            bad += 1
            print("ROUND %d FAILED: %s" % (r, name))
            traceback.print_exc()
    ctx.close()
    ctx0.close()
    print("round %d done" % r, flush=True)
np.random.default_rng = real_rng
print("stress: %d rounds, %d failures" % (rounds, bad))
sys.exit(1 if bad else 0)
