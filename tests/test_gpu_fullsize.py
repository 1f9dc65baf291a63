"""Full BASELINE.json sizes on one MI355X, checked through size-independent
properties (the oracle would need minutes per query here).  The corpus is
generated in HBM with torch and handed to the index by device pointer."""
import numpy as np
import pytest

import support

pytestmark = pytest.mark.gpu

N, D = 10_000_000, 768


@pytest.fixture(scope="module")
def corpus():
    import torch  # first: shares its HIP runtime with the library
    from vettore_amd import nifs
    from bench import build_shard, doc_ids
    dev = torch.device("cuda", 0)
    x = build_shard(torch, dev, N, D, 20260721)
    # a block of 64 identical rows and a few planted near-duplicates of row 12345
    x[5_000_000:5_000_064] = x[5_000_000]
    host_rows = {i: x[i].cpu().numpy() for i in (0, 12345, 5_000_000, 5_000_063, N - 1)}
    return torch, nifs, x, doc_ids, host_rows


def _index(nifs, metric_new, x, doc_ids, start, count):
    ref = metric_new()
    ptr = x.data_ptr() + start * D * 4
    assert nifs.flat_load_device_matrix(ref, doc_ids(start, count), ptr, count, D) == ("ok", ())
    return ref


def _rank_keys(metric, hits):
    import oracle
    return [(support.total_key(oracle.rank_value(metric, np.float32(h[1]))), h[0]) for h in hits]


def test_north_star_size_cosine_properties(corpus, oracle_mod):
    """configs: flat cosine, d=768, N=10M.  Self-hit, reference ordering,
    id tie-break inside a block of identical rows, limit monotonicity, batched ==
    single, quantized search finds planted exact matches."""
    torch, nifs, x, doc_ids, host = corpus
    ref = _index(nifs, nifs.flat_new_cosine, x, doc_ids, 0, N)
    assert len(ref) == N
    for row in (0, 12345, N - 1):
        st, hits = nifs.flat_search(ref, host[row], 10)
        assert st == "ok" and hits[0][0] == b"doc-%d" % (row + 1) and support.close(hits[0][1], 1.0, 1e-5)
        keys = _rank_keys(2, hits)
        assert keys == sorted(keys)
        assert nifs.flat_search(ref, host[row], 5)[1] == hits[:5]
    # 64 identical rows: all tie exactly; the reference order is by id BYTES
    st, hits = nifs.flat_search(ref, host[5_000_000], 70)
    tied = [h for h in hits if np.float32(h[1]).tobytes() == np.float32(hits[0][1]).tobytes()]
    expect = sorted(b"doc-%d" % (i + 1) for i in range(5_000_000, 5_000_064))
    assert [h[0] for h in tied[:64]] == expect
    # batched (MFMA + exact rescoring) == single, bit for bit
    qs = np.stack([host[0], host[12345], host[5_000_000]] + [
        oracle_mod.normalize_l2(np.random.default_rng(s).uniform(-1, 1, D).astype(np.float32)) for s in range(13)])
    st, batch = nifs.flat_search_batch(ref, qs, 10)
    assert st == "ok"
    for i, q in enumerate(qs):
        single = nifs.flat_search(ref, q, 10)[1]
        assert [(h[0], np.float32(h[1]).tobytes()) for h in batch[i]] == \
               [(h[0], np.float32(h[1]).tobytes()) for h in single], i
    # quantized: an exact copy has Hamming distance 0, so it survives the candidate pass
    st, qhits = nifs.flat_quantized_search(ref, host[12345], 100, 10)
    assert st == "ok" and qhits[0][0] == b"doc-12346" and len(qhits) == 10
    keys = _rank_keys(2, qhits)
    assert keys == sorted(keys)
    del ref


def test_config4_shape_l2_shards_merge_to_the_single_index(corpus, oracle_mod):
    """configs[3] on one GPU: the 10M rows as 4 row-block shards (each its own
    index and id-rank space) merged by (rank key, id bytes) must equal the one
    index over all rows -- the property the RCCL merge relies on."""
    torch, nifs, x, doc_ids, host = corpus
    from vettore_amd.sharded import merge_shards
    whole = _index(nifs, nifs.flat_new_l2, x, doc_ids, 0, N)
    rng = np.random.default_rng(4)
    queries = [host[5_000_000], host[N - 1]] + [rng.uniform(-1, 1, D).astype(np.float32) * 0.05 for _ in range(3)]
    singles = [nifs.flat_search(whole, q, 10)[1] for q in queries]
    del whole
    torch.cuda.empty_cache()
    per = N // 4
    shards = [_index(nifs, nifs.flat_new_l2, x, doc_ids, r * per, per) for r in range(4)]
    for q, want in zip(queries, singles):
        parts = [nifs.flat_search_with_keys(s, q, 10)[1] for s in shards]
        got = merge_shards(parts, 10)
        assert [(h[0], np.float32(h[1]).tobytes()) for h in got] == \
               [(h[0], np.float32(h[1]).tobytes()) for h in want]
        keys = _rank_keys(0, got)
        assert keys == sorted(keys)


def test_full_size_results_against_torch_brute_force(corpus, oracle_mod):
    """An independent computation over all 10M rows (torch matmuls in HBM, different
    summation order): the flat hits carry the right scores and are the true top 10 up to
    near-ties; the sign-bit candidate set of quantized_search is the exact Hamming top 100
    including the id tie-break (integer arithmetic: no tolerance)."""
    torch, nifs, x, doc_ids, host = corpus
    ref = _index(nifs, nifs.flat_new_cosine, x, doc_ids, 0, N)
    blob, off = doc_ids(0, N)
    ids = np.array([blob[int(off[i]):int(off[i + 1])] for i in range(N)], dtype="S12")
    order = np.argsort(ids, kind="stable")          # bytewise id order (shorter prefix first)
    id_rank = np.empty(N, dtype=np.int64)
    id_rank[order] = np.arange(N)
    rank_dev = torch.from_numpy(id_rank).to(x.device)
    for seed in (1, 2):
        q = oracle_mod.normalize_l2(np.random.default_rng(seed).uniform(-1, 1, D).astype(np.float32))
        qd = torch.from_numpy(q).to(x.device)
        # --- flat cosine
        st, hits = nifs.flat_search(ref, q, 10)
        assert st == "ok" and len(hits) == 10
        approx = x @ qd
        cand = torch.topk(approx, 2000).indices
        exact = x[cand].double() @ qd.double()       # f64 scores of the best 2000 by the f32 pass
        top = torch.topk(exact, 10)
        true_rows = set(cand[top.indices].tolist())
        kth = float(top.values[-1])
        got_rows = []
        for id_, raw in hits:
            row = int(id_[4:]) - 1
            got_rows.append(row)
            assert support.close(raw, float(x[row].double() @ qd.double()), 1e-5)
        for row in true_rows - set(got_rows):        # only near-ties of the 10th may differ
            assert abs(float(x[row].double() @ qd.double()) - kth) <= 2e-6
        # --- sign-bit Hamming candidates
        st, qhits = nifs.flat_quantized_search(ref, q, 100, 100)
        assert st == "ok" and len(qhits) == 100
        sq = torch.where(qd >= 0, 1.0, -1.0).to(torch.float16)
        keys = torch.empty(N, dtype=torch.int64, device=x.device)
        step = 1_000_000
        for lo in range(0, N, step):
            s = torch.where(x[lo:lo + step] >= 0, 1.0, -1.0).to(torch.float16)
            ham = ((D - (s @ sq).float()) / 2).round().to(torch.int64)   # exact: |sum| <= 768
            keys[lo:lo + step] = ham * (1 << 32) + rank_dev[lo:lo + step]
        want = set(torch.topk(keys, 100, largest=False).indices.tolist())
        assert {int(i[4:]) - 1 for i, _ in qhits} == want
    del ref


@pytest.mark.parametrize("nominate", ["bf16", "f32"])
def test_config3_shape_dot_batch_equals_single_queries(corpus, oracle_mod, nominate):
    """configs[2] at its own shape: metric :dot on UN-normalised rows
    (the bench's `--mode batch` corpus: rows x U(8,24)), N=10M, d=768, ONE batch of 256 on the
    `mfma_scores_kernel<8,...>` instance (nominate = f32: K2) and on `bf16_scores_kernel` (K2b,
    the default: operands rounded to bf16, margin 2^-7 |q| X), where the acceptance bound
    actually bites -- every query's hits equal its own flat_search bit for bit, the number of
    queries the bound could not certify is reported, and a sample is checked against an f64
    brute force over all rows."""
    torch, nifs, x, doc_ids, host = corpus
    g = torch.Generator(device=x.device)
    g.manual_seed(33)
    scale = torch.empty((N, 1), device=x.device).uniform_(8.0, 24.0, generator=g)
    x.mul_(scale)
    try:
        ref = _index(nifs, nifs.flat_new_inner_product, x, doc_ids, 0, N)
        assert nifs.flat_set_batch_nominate(ref, {"f32": 1, "bf16": 2}[nominate]) == "ok"
        qs = np.random.default_rng(20260722).uniform(-1, 1, size=(256, D)).astype(np.float32)
        nifs.flat_set_profiling(ref, True)
        nifs.flat_get_profile(ref, reset=True)
        st, batch = nifs.flat_search_batch(ref, qs, 10)
        assert st == "ok" and len(batch) == 256
        prof = nifs.flat_get_profile(ref, reset=True)
        key = "nominate" if nominate == "bf16" else "batch"
        print("config3 shape (%s): queries=%d fallbacks=%d second passes=%d candidates/query=%.0f" % (
            nominate, prof[key + "_queries"], prof["batch_fallbacks"], prof["nominate_second_passes"],
            prof["nominate_candidates"] / 256))
        assert prof[key + "_queries"] == 256 and prof[key + "_launches"] == 1
        assert prof["batch_fallbacks"] <= 8          # the bound certifies (nearly) every query of this workload
        for i in range(256):
            single = nifs.flat_search(ref, qs[i], 10)[1]
            assert [(h[0], np.float32(h[1]).tobytes()) for h in batch[i]] == \
                   [(h[0], np.float32(h[1]).tobytes()) for h in single], i
        for i in (0, 100, 255):                       # independent f64 scores of the hits, true top-10 up to near-ties
            qd = torch.from_numpy(qs[i]).to(x.device)
            approx = x @ qd
            cand = torch.topk(approx, 2000).indices
            exact = x[cand].double() @ qd.double()
            top = torch.topk(exact, 10)
            true_rows = set(cand[top.indices].tolist())
            kth = float(top.values[-1])
            got_rows = [int(h[0][4:]) - 1 for h in batch[i]]
            for (id_, raw), row in zip(batch[i], got_rows):
                assert support.close(raw, float(x[row].double() @ qd.double()), 1e-5)
            for row in true_rows - set(got_rows):
                assert abs(float(x[row].double() @ qd.double()) - kth) <= 1e-5 * max(1.0, abs(kth))
        if nominate == "bf16":
            # the config as SURVEY 8d writes it -- several groups of 256 in ONE call (four here): consecutive groups
            # alternate between two contexts, group g + 1 queued before group g is waited for.  Every list equals the
            # list its query gets in a call of its own group alone, bit for bit; one pass per group, no fallbacks to speak of
            qs4 = np.concatenate([qs, np.random.default_rng(20260723).uniform(-1, 1, size=(768, D)).astype(np.float32)])
            nifs.flat_get_profile(ref, reset=True)
            st, big = nifs.flat_search_batch(ref, qs4, 10)
            assert st == "ok" and len(big) == 1024
            prof = nifs.flat_get_profile(ref, reset=True)
            assert prof["nominate_launches"] == 4 and prof["nominate_queries"] == 1024 and prof["batch_fallbacks"] <= 16, prof
            bits_of = lambda hits: [(h[0], np.float32(h[1]).tobytes()) for h in hits]
            for i in range(256):
                assert bits_of(big[i]) == bits_of(batch[i]), i
            for g4 in range(1, 4):
                st, part = nifs.flat_search_batch(ref, qs4[256 * g4:256 * (g4 + 1)], 10)
                assert st == "ok"
                for i in range(256):
                    assert bits_of(big[256 * g4 + i]) == bits_of(part[i]), (g4, i)
        del ref
    finally:
        x.div_(scale)   # the fixture is shared (module scope); later tests see (almost) the same rows again
