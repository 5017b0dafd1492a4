"""Property tests (hypothesis) of the IM arithmetic on the oracle -- the invariants SURVEY section 4 lists: final and IM
are disjoint, sizes add up, N = 2 binary IM is the XOR of the votes, permutation invariance over models (binary),
model-0 anchoring is irrelevant when all models agree (multiclass)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import im_oracle as O

shapes = st.tuples(st.integers(1, 4), st.integers(1, 12), st.integers(1, 12))     # N models, H, W


def _probs(seed, n, h, w, k):
    rng = np.random.default_rng(seed)
    p = rng.random((n, h, w, k), dtype=np.float32)
    p[rng.random((n, h, w, k)) < 0.15] = 0.5                                       # exactly on the threshold
    return p


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 10 ** 6), shapes, st.booleans())
def test_binary_invariants(seed, nhw, ge):
    n, h, w = nhw
    p = _probs(seed, n, h, w, 1)
    r = O.im_binary(p, 0.5, ge)
    final, im = r["final"][0], r["im"]
    votes = O.threshold_votes(p[..., 0], 0.5, ge)
    s = votes.sum(0)
    assert set(np.unique(final)) <= {0, 255} and set(np.unique(im)) <= {0, 255}
    assert not np.any((final > 0) & (im > 0))                                      # disjoint
    assert int(r["pred_size"]) == int((s == n).sum()) and int(r["im_size"]) == int(((s > 0) & (s < n)).sum())
    assert int(r["pred_size"]) + int(r["im_size"]) + int((s == 0).sum()) == h * w  # sizes add up
    if n == 2:
        assert np.array_equal(im > 0, votes[0] != votes[1])                        # XOR
    if n == 1:
        assert not im.any()
    perm = np.random.default_rng(seed + 1).permutation(n)
    r2 = O.im_binary(p[perm], 0.5, ge)                                             # models are exchangeable
    assert np.array_equal(r2["final"][0], final) and np.array_equal(r2["im"], im)
    assert int(r2["im_size"]) == int(r["im_size"]) and int(r2["pred_size"]) == int(r["pred_size"])


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 10 ** 6), shapes, st.integers(2, 9))
def test_multiclass_invariants(seed, nhw, k):
    n, h, w = nhw
    p = _probs(seed, n, h, w, k)
    r = O.im_multiclass(p)
    labels = O.argmax_first(p)
    agree = np.all(labels == labels[0:1], axis=0)
    assert np.array_equal(r["im"] > 0, ~agree)
    assert int(r["im_size"]) == int((~agree).sum())
    assert np.array_equal(r["final"], np.where(agree, labels[0], 0).astype(np.uint8))   # class ids, 0 where inconsistent
    assert not np.any((r["final"] > 0) & (r["im"] > 0))
    # when every model agrees everywhere, which model is the anchor does not matter
    same = np.repeat(p[:1], n, axis=0)
    ra = O.im_multiclass(same)
    assert not ra["im"].any() and np.array_equal(ra["final"], labels[0].astype(np.uint8))
    # ties resolve to the lowest class index
    tie = p.copy()
    tie[..., 0] = tie.max(-1)
    assert np.all(O.argmax_first(tie) == 0)


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 10 ** 6), st.integers(1, 3), st.integers(2, 10), st.integers(2, 10))
def test_hela_combination(seed, n, h, w):
    p = _probs(seed, n, h, w, 3)
    r = O.im_binary(p, 0.5, True)                                                   # HeLa: >= and three channels
    per = [O.im_binary(p[..., c:c + 1], 0.5, True) for c in range(3)]
    assert np.array_equal(r["im"], np.maximum.reduce([q["im"] for q in per]))      # combined IM = max
    assert int(np.sum(r["im_size"])) == sum(int(np.sum(q["im_size"])) for q in per)     # sizes: overlaps counted per channel
