"""Drop-in boundary behaviour on the GPU (-m gpu): the address-keyed SRS cache follows the CONTENTS of the caller's memory,
asynchronous entries on different streams do not share intermediate data, NULL means the default stream for the transforms,
and the one exchange step of the multi-GPU MSM runs over RCCL (backend nccl) even on one GPU."""
import os

import numpy as np
import pytest

from oracle.pyoracle import FR_MODULUS, aligned_copy, aligned_empty
from tests.util import NTT_SEED, SCALAR_SEED, SRS_SEED, limbs, noncanonical

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from barretenberg_amd import BbGpu
    os.environ["BBGPU_SRS_CACHE_BYTES"] = str(12 << 20)  # room for two 2048-point tables with their window tables (4.9 MiB each)
    g = BbGpu(device=0)
    g.set_host_thresholds(0, 0)
    yield g
    g.shutdown()
    del os.environ["BBGPU_SRS_CACHE_BYTES"]


@pytest.fixture(scope="module")
def tables(oracle):
    """three different 2048-point SRS (different secrets) as endo tables, a 24-point one, and scalars"""
    out = []
    for k in range(3):
        x = oracle.random_scalars(SRS_SEED + 17 * k, 1)[0]
        out.append(oracle.point_table(oracle.make_srs(x, 2048)))
    small = oracle.point_table(oracle.make_srs(oracle.random_scalars(SRS_SEED + 99, 1)[0], 24))
    return out, small, oracle.random_scalars(SCALAR_SEED, 2048)


def test_srs_cache_follows_buffer_contents(gpu, oracle, tables):
    """a table registered on first sight must not be served after its memory holds other points --
    refilled in place (n >= 1024), replaced by a different-size table at the same address, or a small table at an interior address"""
    (A, B, C), small, sc = tables
    n = 2048
    buf = aligned_empty((2 * n, 8))
    buf[:] = A
    assert np.array_equal(gpu.pippenger(sc, buf, n)[:8], oracle.msm_affine(sc, A, n)[:8])
    live0, auto0, _ = gpu.srs_cache_stats()
    assert auto0 >= 1
    buf[:] = B                                     # same address, same size, other points
    assert np.array_equal(gpu.pippenger(sc, buf, n)[:8], oracle.msm_affine(sc, B, n)[:8])
    assert gpu.srs_cache_stats()[1] == auto0       # the stale entry was evicted, not leaked
    buf[:2 * 1500] = C[:2 * 1500]                  # a different-size table at the same address (1500 points)
    assert np.array_equal(gpu.pippenger(sc, buf, 1500)[:8], oracle.msm_affine(sc, aligned_copy(C[:3000]), 1500)[:8])
    assert np.array_equal(gpu.pippenger(sc, buf, 1500)[:8], oracle.msm_affine(sc, aligned_copy(C[:3000]), 1500)[:8])  # now served from the cache
    buf[400:448] = small                           # the verifier's freshly built 24 points at an interior address of a cached range
    got = gpu.pippenger(aligned_copy(sc[:24]), buf[400:], 24)
    assert np.array_equal(got[:8], oracle.msm_affine(aligned_copy(sc[:24]), small, 24)[:8])
    # sub-slices of an unchanged cached table are still served (scalar_multiplication.cpp:720-726)
    buf[:] = A
    gpu.pippenger(sc, buf, n)
    want = oracle.msm_affine(aligned_copy(sc[:300]), aligned_copy(A[200:800]), 300)
    assert np.array_equal(gpu.pippenger(aligned_copy(sc[:300]), buf[200:], 300)[:8], want[:8])
    # batched entry: same rule
    buf[:] = B
    outs = gpu.batched_scalar_multiplications([(buf, aligned_copy(sc[o:o + 1024]), 1024) for o in (0, 1024)])
    for o, out in zip((0, 1024), outs):
        assert np.array_equal(out[:8], oracle.msm_affine(aligned_copy(sc[o:o + 1024]), aligned_copy(B[:2048]), 1024)[:8])


def test_explicit_registration_is_not_served_stale(gpu, oracle, tables):
    (A, B, C), small, sc = tables
    import torch
    n = 2048
    buf = aligned_empty((2 * n, 8))
    buf[:] = C
    h = gpu.srs_register(buf)
    want_c = oracle.msm_affine(sc, C, n)
    assert np.array_equal(gpu.pippenger(sc, buf, n)[:8], want_c[:8])
    buf[:] = A                                     # mutated in place without bbgpu_srs_release (documented misuse)
    assert np.array_equal(gpu.pippenger(sc, buf, n)[:8], oracle.msm_affine(sc, A, n)[:8])  # host-pointer calls follow the memory
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    assert np.array_equal(gpu.msm_device(h, d.data_ptr(), n)[:8], want_c[:8])              # the handle still names what was registered
    gpu.srs_release(h)


def test_srs_cache_lru_under_byte_cap(gpu, oracle, tables):
    (A, B, C), small, sc = tables
    n = 2048
    bufs = [aligned_copy(t) for t in (A, B, C)]
    for b, t in zip(bufs, (A, B, C)):
        assert np.array_equal(gpu.pippenger(sc, b, n)[:8], oracle.msm_affine(sc, t, n)[:8])
        live, auto, held = gpu.srs_cache_stats()
        assert held <= (12 << 20), held
    assert gpu.srs_cache_stats()[1] <= 2           # three tables seen, at most two fit the cap
    for b, t in zip(bufs, (A, B, C)):              # evicted tables come back on demand
        assert np.array_equal(gpu.pippenger(sc, b, n)[:8], oracle.msm_affine(sc, t, n)[:8])


def test_srs_cache_catches_a_rewritten_middle_slice(gpu, oracle, tables):
    """a table refilled in place only in the MIDDLE (first and last row unchanged) escaped the fixed 16-row sample for ever;
    the sampled rows now move on with every check, so the stale copy is dropped within a few calls -- and the registry does not grow by one
    entry per re-registration"""
    (A, B, C), small, sc = tables
    n = 2048
    buf = aligned_empty((2 * n, 8))
    buf[:] = A
    want_a = oracle.msm_affine(sc, A, n)
    assert np.array_equal(gpu.pippenger(sc, buf, n)[:8], want_a[:8])
    mixed = A.copy()
    mixed[2 * 300:2 * 1700] = B[2 * 300:2 * 1700]  # 68 % of the rows replaced, both ends kept
    want_m = oracle.msm_affine(sc, mixed, n)
    buf[:] = mixed
    seen = [bool(np.array_equal(gpu.pippenger(sc, buf, n)[:8], want_m[:8])) for _ in range(4)]
    assert seen[-1] and all(seen[seen.index(True):]), seen  # caught (each check misses with probability 0.32^14), and stays correct afterwards
    live0 = gpu.srs_cache_stats()[0]
    for t in (A, mixed, A, mixed, A):                 # five more evict / re-register rounds
        buf[:] = t
        gpu.pippenger(sc, buf, n)
    assert gpu.srs_cache_stats()[0] == live0


def test_exact_cache_mode_sees_one_rewritten_point_on_the_very_next_call(gpu, oracle, golden):
    """the reference reads the caller's points on every call (scalar_multiplication.cpp:604-617); the address-keyed cache answers from a
    resident copy after a 16-row sample, so ONE point rewritten in the middle of a cached 2^16-point table is served stale (almost) for ever.  In exact
    mode (bbgpu_srs_set_validate / BBGPU_SRS_VALIDATE=full) the very next call returns the point the REFERENCE computed for the rewritten table
    (tests/golden/msm_r5.json `rewrite`, tools/gen_golden_r5.py) -- through pippenger(), through the batched entry, for a table registered on first sight
    and for an explicitly registered one (whose handle keeps naming what was registered)."""
    import torch
    from tests.util import sha
    g = golden("msm_r5.json")
    rw = g["rewrite"]
    n, at, src = rw["n"], rw["index"], rw["takes_point"]
    h0, table = gpu.srs_generate(limbs(g["srs_secret_mont"]), src + 1, True)  # 2^16 + 6 points of the fixtures' SRS (digest pinned over 2^21 by the parity tests)
    gpu.srs_release(h0)
    scalars = oracle.random_scalars(SCALAR_SEED, n)

    def same(out, case):
        return np.array_equal(out[0:4], limbs(case["x"])) and np.array_equal(out[4:8], limbs(case["y"]))
    try:  # (a table beyond the module's 12 MiB cache cap is still kept: everything else registered on first sight is evicted for it)
        # sampled mode (the default): the stale copy survives the next call -- the window the exact mode closes (16 of 65536 rows sampled)
        buf = aligned_copy(table[:2 * n])
        assert same(gpu.pippenger(scalars, buf, n), rw["before"])
        buf[2 * at:2 * at + 2] = table[2 * src:2 * src + 2]
        assert same(gpu.pippenger(scalars, buf, n), rw["before"])  # stale: identical inputs, not the reference's output
        # exact mode as the default for tables seen from now on
        gpu.srs_set_validate(-1, True)
        buf = aligned_copy(table[:2 * n])  # a new address: registered on first sight, in exact mode
        assert same(gpu.pippenger(scalars, buf, n), rw["before"])
        assert same(gpu.pippenger(scalars, buf, n), rw["before"])  # served from the copy (full check passes)
        buf[2 * at:2 * at + 2] = table[2 * src:2 * src + 2]
        assert same(gpu.pippenger(scalars, buf, n), rw["after"])   # the very next call
        assert same(gpu.pippenger(scalars, buf, n), rw["after"])
        buf[2 * at:2 * at + 2] = table[2 * at:2 * at + 2]          # and back, through the batched entry (two jobs over the same range: one check)
        outs = gpu.batched_scalar_multiplications([(buf, scalars, n), (buf, scalars, n)])
        assert same(outs[0], rw["before"]) and same(outs[1], rw["before"])
        buf[2 * at:2 * at + 2] = table[2 * src:2 * src + 2]
        outs = gpu.batched_scalar_multiplications([(buf, scalars, n), (buf, scalars, n)])
        assert same(outs[0], rw["after"]) and same(outs[1], rw["after"])
        # sub-slices of the (re-uploaded) table are served as before
        want = oracle.msm_affine(aligned_copy(scalars[:5000]), aligned_copy(buf[2 * (at - 100):2 * (at + 4900)]), 5000)
        assert np.array_equal(gpu.pippenger(aligned_copy(scalars[:5000]), buf[2 * (at - 100):], 5000)[:8], want[:8])
        # a table large enough that the call's scalars bypass the staging pool (2^19 points: 16 MiB): there the full check runs in the BACKGROUND on the pool's
        # helper threads, from before the upload to after the launches.  Checked against the resident path on fresh registrations of the same contents
        # (the arithmetic is pinned elsewhere; this pins WHICH contents the call used).
        big_n = 1 << 19
        hb, big = gpu.srs_generate(limbs(g["srs_secret_mont"]), big_n + 8, True)
        gpu.srs_release(hb)
        big_sc = oracle.random_scalars(SCALAR_SEED + 5, big_n)
        d_big = torch.from_numpy(big_sc.view(np.int64)).cuda()

        def resident(tab):
            hh = gpu.srs_register(aligned_copy(tab))
            gpu.srs_set_validate(hh, False)
            out = gpu.msm_device(hh, d_big.data_ptr(), big_n)
            gpu.srs_release(hh)
            return out
        bufb = aligned_copy(big[:2 * big_n])
        want0 = resident(bufb)
        assert np.array_equal(gpu.pippenger(big_sc, bufb, big_n), want0)      # registered on first sight, exact mode
        assert np.array_equal(gpu.pippenger(big_sc, bufb, big_n), want0)      # served from the copy, background check passes
        mid = big_n // 2 + 12345
        bufb[2 * mid:2 * mid + 2] = big[2 * (big_n + 3):2 * (big_n + 3) + 2]   # ONE point rewritten in the middle
        want1 = resident(bufb)
        assert not np.array_equal(want1, want0)
        assert np.array_equal(gpu.pippenger(big_sc, bufb, big_n), want1)      # the very next call
        assert np.array_equal(gpu.pippenger(big_sc, bufb, big_n), want1)
        gpu.srs_set_validate(-1, False)
        # per-handle flag on an explicitly registered table
        buf2 = aligned_copy(table[:2 * n])
        h = gpu.srs_register(buf2)
        gpu.srs_set_validate(h, True)
        assert same(gpu.pippenger(scalars, buf2, n), rw["before"])
        buf2[2 * at:2 * at + 2] = table[2 * src:2 * src + 2]
        assert same(gpu.pippenger(scalars, buf2, n), rw["after"])
        d = torch.from_numpy(scalars.view(np.int64)).cuda()
        assert same(gpu.msm_device(h, d.data_ptr(), n), rw["before"])  # the handle still names what was registered
        gpu.srs_release(h)
    finally:
        gpu.srs_set_validate(-1, False)


def test_two_transforms_in_flight_on_two_streams(gpu, oracle):
    """back-to-back transforms on different streams used to share one pass-1 scratch buffer"""
    import torch
    n = 1 << 16
    xs = [noncanonical(oracle.random_scalars(NTT_SEED + 31 * k, n), FR_MODULUS) for k in range(4)]
    want = [oracle.ntt(x, "fft") for x in xs]
    streams = [torch.cuda.Stream() for _ in range(4)]
    ds = [torch.from_numpy(x.view(np.int64)).cuda() for x in xs]
    torch.cuda.synchronize()
    for _ in range(3):  # several rounds, no synchronisation in between
        for d, s in zip(ds, streams):
            gpu.ntt_device(d.data_ptr(), n, "fft", stream=s.cuda_stream)
        for d, s in zip(ds, streams):
            gpu.ntt_device(d.data_ptr(), n, "ifft", stream=s.cuda_stream)
    for d, s in zip(ds, streams):
        gpu.ntt_device(d.data_ptr(), n, "fft", stream=s.cuda_stream)
    torch.cuda.synchronize()
    for d, w in zip(ds, want):
        assert np.array_equal(d.cpu().numpy().view(np.uint64), w)


def test_null_stream_is_the_default_stream(gpu, oracle):
    """NULL = the legacy default stream: work the caller queued there is ordered with the transform"""
    import torch
    n = 1 << 18
    x = oracle.random_scalars(NTT_SEED + 5, n)
    want = oracle.ntt(x, "coset_fft")
    d = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    h = torch.from_numpy(x.view(np.int64)).pin_memory()
    for _ in range(3):
        d.zero_()
        d.copy_(h, non_blocking=True)               # producer on the default stream, not waited for
        gpu.ntt_device(d.data_ptr(), n, "coset_fft")  # stream=None
        out = d.clone()                              # consumer on the default stream
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want)


def test_rank_share_of_the_window_tables(gpu, oracle, tables):
    """multi-GPU memory layout: a rank keeps only the digit windows its 1/N share of the (window, point) rows touches (bbgpu_set_table_share);
    the eight shares, each against its own restricted table, still fold to the MSM, and windows outside the share are refused"""
    import torch
    from barretenberg_amd import BbGpuError
    (A, B, C), small, sc = tables
    n = 2048
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    want = oracle.msm_affine(sc, A, n)
    for world in (2, 8):
        parts = []
        for r in range(world):
            gpu.set_table_share(r, world)
            h = gpu.srs_register(aligned_copy(A))
            gpu.set_table_share(0, 1)
            W = gpu.srs_num_windows(h, n)
            R = W * n
            parts.append(gpu.msm_wait(gpu.msm_device_rows_async(h, d.data_ptr(), n, R * r // world, R * (r + 1) // world)))
            if r == 0 and world == 8:
                with pytest.raises(BbGpuError, match="keeps"):
                    gpu.msm_device(h, d.data_ptr(), n)  # the full window range is not resident on this rank
            gpu.srs_release(h)
        assert np.array_equal(gpu.g1_sum(np.stack(parts))[:8], want[:8]), world


def test_point_range_slices_against_the_oracle(gpu, oracle, tables):
    """the other multi-GPU split (bbgpu_set_point_share, bench.py --shard points): rank r registers points [n r / N, n (r + 1) / N) of the table as its own SRS and
    runs the ordinary MSM over the matching scalars; the N results (uneven ranges, up to four in flight) fold to the oracle's point.  The call only moves the
    window size of the slice's tables: slices of 2^17 / 2^18 points of a 2^20-point MSM take 16 / 17 bits (16 / 15 windows), alone they take 15 bits (17 windows)"""
    import torch
    (A, B, C), small, sc = tables
    n = 2048
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    want = oracle.msm_affine(sc, A, n)
    for world in (2, 3, 8):
        cuts = [n * r // world for r in range(world + 1)]
        gpu.set_point_share(world)
        hs = [gpu.srs_register(aligned_copy(A[2 * a:2 * b])) for a, b in zip(cuts[:-1], cuts[1:])]
        gpu.set_point_share(1)
        tickets, parts = [], []
        for r, h in enumerate(hs):
            tickets.append(gpu.msm_device_async(h, d.data_ptr() + cuts[r] * 32, cuts[r + 1] - cuts[r]))
            if len(tickets) == 4:
                parts.append(gpu.msm_wait(tickets.pop(0)))
        parts += [gpu.msm_wait(t) for t in tickets]
        assert np.array_equal(gpu.g1_sum(np.stack(parts))[:8], want[:8]), world
        for h in hs:
            gpu.srs_release(h)
    probe = gpu.srs_register(aligned_copy(A))
    with_tables = gpu.srs_has_window_tables(probe)
    gpu.srs_release(probe)
    if with_tables:  # the suite's pass without tables has no window size to look at
        x = np.array([3, 0, 0, 0], dtype=np.uint64)
        for world, m, windows in ((8, 1 << 17, 16), (4, 1 << 18, 15), (1, 1 << 17, 17)):
            gpu.set_point_share(world)
            h = gpu.srs_generate(x, m, first=5)
            gpu.set_point_share(1)
            assert gpu.srs_num_windows(h, m) == windows, (world, m)
            gpu.srs_release(h)


def test_partial_sum_exchange_over_rccl_world_size_1(gpu, oracle, tables):
    """the multi-GPU MSM's one exchange step over backend nccl (= RCCL) with the ranks this box has: init, all_gather of the
    96-byte partial sums on the GPU, identical fold -- the same objects bench.py uses for N > 1 (barretenberg_amd/sharding.py)"""
    import torch
    import torch.distributed as dist
    from barretenberg_amd.sharding import PartialSumExchange, pipelined_steps
    (A, B, C), small, sc = tables
    n = 2048
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        h = gpu.srs_register(aligned_copy(A))
        d = torch.from_numpy(sc.view(np.int64)).cuda()
        ex = PartialSumExchange(gpu, 1, torch.device("cuda:0"))
        res = pipelined_steps(3, lambda: gpu.msm_device_async(h, d.data_ptr(), n), gpu.msm_wait, ex)
        want = oracle.msm_affine(sc, A, n)
        assert len(res) == 3 and all(np.array_equal(r[:8], want[:8]) for r in res)
        gpu.srs_release(h)
    finally:
        dist.destroy_process_group()


def test_concurrent_callers_of_the_drop_in_entries(gpu, oracle, golden):
    """The reference enters pippenger() from inside `#pragma omp parallel for` (scalar_multiplication.cpp:731-738) and has no error channel: eight
    host threads call bbgpu_msm_g1 / bbgpu_ntt on their own buffers at once while the main thread HOLDS an asynchronous ticket (and one of them
    keeps issuing and collecting tickets of its own).  Calls are serialised by the library mutex and take whatever MSM slots are free -- none may
    fail with BBGPU_ERR_STATE, every result equals the oracle's."""
    import threading
    import torch
    g = golden("msm.json")
    n_big = 65536
    x = limbs(g["srs_secret_mont"])
    h, table = gpu.srs_generate(x, n_big, True)
    scalars = oracle.random_scalars(SCALAR_SEED, n_big)
    d_sc = torch.from_numpy(scalars.view(np.int64)).cuda()
    want_big = [c for c in g["cases"] if c["n"] == n_big and "x" in c][0]
    sizes = [4096, 1000, 10000, 65536, 100, 4096, 16, 65536]
    want = {n: [c for c in g["cases"] if c["n"] == n and "x" in c][0] for n in set(sizes)}
    ntt_in = {k: noncanonical(oracle.random_scalars(NTT_SEED + 300 + k, 1 << (8 + k)), FR_MODULUS) for k in range(4)}
    ntt_want = {k: oracle.ntt(v, "coset_fft") for k, v in ntt_in.items()}
    held = gpu.msm_device_async(h, d_sc.data_ptr(), n_big)  # stays in flight for the whole test
    errors, rounds = [], 6

    def worker(i):
        try:
            n = sizes[i]
            sc = aligned_copy(scalars[:n])
            for r in range(rounds):
                out = gpu.pippenger(sc, table, n)
                assert np.array_equal(out[0:4], limbs(want[n]["x"])) and np.array_equal(out[4:8], limbs(want[n]["y"])), ("msm", i, r)
                k = (i + r) % 4
                got = gpu.coset_fft(ntt_in[k].copy())
                assert np.array_equal(got, ntt_want[k]), ("ntt", i, r)
                if i == 0:  # tickets of its own beside the held one
                    t = gpu.msm_device_async(h, d_sc.data_ptr(), 4096)
                    o2 = gpu.msm_wait(t)
                    assert np.array_equal(o2[0:4], limbs(want[4096]["x"])), ("ticket", r)
        except BaseException as exc:  # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    out = gpu.msm_wait(held)
    assert not errors, errors
    assert np.array_equal(out[0:4], limbs(want_big["x"])) and np.array_equal(out[4:8], limbs(want_big["y"]))
    gpu.srs_release(h)


def test_transform_tables_stay_under_their_byte_budget(gpu, oracle, golden):
    """the twiddle / twist tables are built per domain size (128 MiB at 2^20, 512 MiB at 2^22) and used to stay for the life of
    the process.  Under BBGPU_NTT_TABLE_BYTES = 1 GiB a walk through the domains 2^10 .. 2^24 (twice: the second pass rebuilds what the first
    evicted) keeps bbgpu_memory_stats().ntt_table_bytes under the budget, and every output still equals the oracle's (to 2^16), the reference's
    digests (2^20, 2^22) or survives the round trip (the sizes between and 2^23 / 2^24, three-pass transforms holding two table sets at once)."""
    from tests.util import sha
    g = golden("ntt.json")
    const = limbs(g["constant"])
    cap = 1 << 30
    os.environ["BBGPU_NTT_TABLE_BYTES"] = str(cap)
    try:
        seen_sets = []
        for sweep in range(2):
            for lg in list(range(10, 25)):
                n = 1 << lg
                if lg <= 16:
                    co = noncanonical(oracle.random_scalars(NTT_SEED + lg, n), FR_MODULUS)
                    for kind in ("fft", "coset_ifft"):
                        assert np.array_equal(gpu.ntt(co.copy(), kind), oracle.ntt(co, kind)), (sweep, lg, kind)
                elif lg in (20, 22):
                    co = noncanonical(oracle.random_scalars(NTT_SEED, n), FR_MODULUS)
                    for case in [x for x in g["large"] if x["n"] == n and x["kind"] in ("fft", "coset_fft_with_constant")]:
                        assert sha(gpu.ntt(co.copy(), case["kind"], const)) == case["sha256"], (sweep, lg, case["kind"])
                else:
                    x = oracle.random_scalars(NTT_SEED + 900 + lg, n)
                    assert np.array_equal(gpu.coset_ifft(gpu.coset_fft(x.copy())), x), (sweep, lg)
                m = gpu.memory_stats()
                assert m["ntt_table_cap_bytes"] == cap and m["ntt_table_bytes"] <= cap, (sweep, lg, m)
                seen_sets.append(m["ntt_table_sets"])
        assert max(seen_sets) < 15 and seen_sets[-1] >= 1  # something was evicted on the way: 15 domain sizes were walked (plus the row domains of 2^23 / 2^24)
    finally:
        del os.environ["BBGPU_NTT_TABLE_BYTES"]
    m = gpu.memory_stats()
    assert m["msm_workspace_bytes"] >= 0 and m["staging_bytes"] > 0 and m["pinned_host_bytes"] > 0


def test_reference_calling_pattern_openmp_pippenger(tmp_path):
    """scalar_multiplication.cpp:703-738 as the reference does it, through the shim's mangled symbols: OpenMP threads call pippenger() on sub-slices of one
    resident point table at the same time (and fft / ifft on buffers of their own); their partial sums add up to the one-call result.  8 threads at
    2^16 points, 5 at 100,003 (ragged ranges).  BBGPU_SHIM_STRICT=1: no host answer may stand in for a kernel."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "barretenberg_amd")
    exe = str(tmp_path / "test_shim_omp")
    subprocess.run(["g++", "-std=c++17", "-O2", "-fopenmp", "-Wno-invalid-offsetof", "-o", exe, os.path.join(root, "tests", "cpp", "test_shim_omp.cpp"), "-L" + pkg, "-lbbshim", "-lbbgpu",
                    "-Wl,-rpath," + pkg], check=True)
    for n, threads in ((1 << 16, 8), (100003, 5)):
        r = subprocess.run([exe, str(n), str(threads)], capture_output=True, text=True, timeout=300, env=dict(os.environ, BBGPU_SHIM_STRICT="1"))
        assert r.returncode == 0 and "ALL OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
