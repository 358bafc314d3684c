"""Shared checkers of the test-suite (test infrastructure only)."""
import numpy as np


def assert_mined_negatives_contract(neg, neg_ref, scores, num_pos, negpos_ratio=3, ulps=2):
    """Hard-negative mining contract (multibox_loss.py:93-106).

    The mining score ``loss_c_all = log(sum(exp(x - max))) + max - x[target]`` passes through exp / log, which the reference
    evaluates with whatever 1-ulp approximation its backend ships (SLEEF on the CPU, libdevice on a GPU) -- its own choice among
    priors whose scores tie to the last bits AT the cut-off is platform dependent.  So: the number of mined negatives per image
    is exact, and the mined set equals the reference's except for priors whose score lies within ``ulps`` ulp of the image's
    cut-off score (the k-th largest, k = min(ratio * n_pos, P - 1)).  ``scores`` are fp32 mining scores [B, P] (positives = 0)."""
    neg, neg_ref = np.asarray(neg, bool), np.asarray(neg_ref, bool)
    scores = np.asarray(scores, np.float32)
    B, P = neg.shape
    assert np.array_equal(neg.sum(1), neg_ref.sum(1)), 'number of mined negatives differs'
    k = np.minimum(negpos_ratio * np.asarray(num_pos).reshape(B), P - 1)
    assert np.array_equal(neg.sum(1), k)
    for b in range(B):
        bad = np.nonzero(neg[b] != neg_ref[b])[0]
        if bad.size == 0 or k[b] == 0:
            continue
        cut = np.sort(scores[b])[::-1][k[b] - 1]
        tol = ulps * np.spacing(np.float32(abs(cut)))
        assert np.all(np.abs(scores[b, bad] - cut) <= tol), (b, bad, scores[b, bad], cut)
