"""The library is re-entrant per HIP stream (threading): independent contexts on their own streams,
driven from their own host threads at the same time, give the same bits as the same work done alone."""
import threading

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def test_contexts_on_concurrent_streams_and_threads(oracle):
    import torch
    import libxaac_amd
    n_threads, n, rounds = 4, 512, 12
    rng = np.random.default_rng(99)
    work = []
    for t in range(n_threads):
        spec, ovl = oracle_lib.random_case(rng, n, mag=12 + 4 * t, ovl_mag=10 + t)
        ics = np.stack([rng.integers(0, 4, n), rng.integers(0, 2, n)], 1).astype(np.uint8)
        state = np.stack([rng.integers(0, 4, n), rng.integers(0, 2, n)], 1).astype(np.uint8)
        work.append((spec, ics, ovl, state))
    # what each thread's chain of `rounds` frames (same spectrum every frame, state carried) has to produce
    want = []
    for spec, ics, ovl, state in work:
        o, s = ovl, state
        for _ in range(rounds):
            r = oracle.imdct_batch(spec, ics, o, s, ch_fac=2)
            o, s = r["overlap"], r["state"]
        want.append(r)
    got = [None] * n_threads
    errors = []

    def run(t):
        try:
            stream = torch.cuda.Stream()
            ctx = libxaac_amd.XaacContext(0, stream.cuda_stream)
            spec, ics, ovl, state = work[t]
            with torch.cuda.stream(stream):
                d_spec, d_ics = torch.from_numpy(spec).cuda(), torch.from_numpy(ics).cuda()
                d_ovl, d_state = torch.from_numpy(ovl.copy()).cuda(), torch.from_numpy(state.copy()).cuda()
                pcm = torch.zeros(n * 1024, dtype=torch.int16, device="cuda")
                out32 = torch.zeros(n * 1024, dtype=torch.int32, device="cuda")
                stream.synchronize()
                for _ in range(rounds):
                    ctx.imdct_process_batch(d_spec, d_ics, d_ovl, d_state, out32=out32, pcm16=pcm, ch_fac=2)
                ctx.sync()
                got[t] = (pcm.cpu().numpy(), out32.cpu().numpy(), d_ovl.cpu().numpy(), d_state.cpu().numpy())
            ctx.close()
        except Exception as e:  # surfaced below: an exception in a thread must fail the test
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for t in range(n_threads):
        pcm, out32, ovl, state = got[t]
        assert np.array_equal(pcm.reshape(n, 1024), want[t]["pcm16"]), t
        assert np.array_equal(out32.reshape(n, 1024), want[t]["out32"]), t
        assert np.array_equal(ovl, want[t]["overlap"]) and np.array_equal(state, want[t]["state"]), t
