"""Heat-map decode: oracle vs reference-generated goldens on CPU, HIP kernel vs oracle on GPU."""
import numpy as np
import pytest
import torch

from oracle import otpose_oracle as O
from tests.conftest import seeded


def test_oracle_matches_reference_golden(golden):
    g = golden("decode")
    hm = g["hm"].numpy()
    p, m = O.get_max_preds(hm)
    assert np.array_equal(p, g["max_preds"].numpy()) and np.array_equal(m, g["maxvals"].numpy())
    p, m = O.get_final_preds(hm)
    assert np.array_equal(p, g["final_preds"].numpy())
    # the cases the fixture was built for
    assert p[0, 0].tolist() == [0.0, 0.0]                    # max <= 0 -> masked
    assert g["max_preds"][0, 1].tolist() == [7.0, 5.0]       # first of two equal maxima
    assert p[1, 2].tolist() == [0.0, 0.0]                    # corner peak, no shift
    assert abs(p[1, 3, 0] - 2.0) == 0.25                     # strict 1 < px passes at px = 2
    assert p[1, 4, 1] == 1.0                                 # py = 1 fails the test: y not shifted


def test_oracle_similarity_transform_is_the_three_point_affine():
    """transform_preds for rot = 0: solve the affine through the three points of utils/transform.py:76-105 by least
    squares and compare with the closed-form similarity used by the oracle and the kernel."""
    w, h = 18, 24
    center, scale = np.array([123.5, 77.25], np.float32), np.array([1.3, 1.7], np.float32)
    src_w = scale[0] * 200.0
    dst = np.array([[w * .5, h * .5], [w * .5, h * .5 - w * .5]], np.float32)
    src = np.array([center, center + [0, -src_w * .5]], np.float32)
    third = lambda a, b: b + np.array([-(a - b)[1], (a - b)[0]], np.float32)     # noqa: E731
    dst = np.vstack([dst, third(dst[0], dst[1])])
    src = np.vstack([src, third(src[0], src[1])])
    A = np.hstack([dst, np.ones((3, 1), np.float32)])
    T = np.linalg.solve(A.astype(np.float64), src.astype(np.float64)).T          # 2x3 affine dst -> src
    hm = np.zeros((1, 2, h, w), np.float32)
    hm[0, 0, 10, 4] = 1.0
    hm[0, 1, 20, 15] = 2.0
    p, _ = O.get_final_preds(hm, center[None], scale[None])
    raw, _ = O.get_final_preds(hm)
    for j in range(2):
        want = T @ np.array([raw[0, j, 0], raw[0, j, 1], 1.0])
        assert np.allclose(p[0, j], want, atol=1e-3)


@pytest.mark.gpu
def test_hip_decode_matches_oracle(golden):
    from otpose_amd import ops
    g = golden("decode")
    hm = g["hm"]
    p, m = ops.get_max_preds(hm.cuda())
    assert torch.equal(p.cpu(), g["max_preds"]) and torch.equal(m.cpu(), g["maxvals"])
    p, m = ops.get_final_preds(hm.cuda())
    assert torch.equal(p.cpu(), g["final_preds"])
    center = seeded((3, 2), 5, 50.0) + 200.0
    scale = seeded((3, 2), 6, 0.2).abs() + 1.0
    p, _ = ops.get_final_preds(hm.cuda(), center.cuda(), scale.cuda())
    ref, _ = O.get_final_preds(hm.numpy(), center.numpy(), scale.numpy())
    assert np.allclose(p.cpu().numpy(), ref, atol=1e-3)


@pytest.mark.gpu
def test_hip_decode_full_size_and_nan():
    from otpose_amd import ops
    hm = seeded((16, 17, 96, 72), 9)
    p, m = ops.get_final_preds(hm.cuda())
    ref_p, ref_m = O.get_final_preds(hm.numpy())
    assert np.array_equal(p.cpu().numpy(), ref_p) and np.array_equal(m.cpu().numpy(), ref_m)
    hm[3, 4, 50, 60] = float("nan")                          # numpy: the first NaN is the arg-maximum, max is NaN
    p, m = ops.get_max_preds(hm.cuda())
    assert torch.isnan(m[3, 4, 0]) and p[3, 4].tolist() == [0.0, 0.0]
    ref_p, _ = O.get_max_preds(hm.numpy())
    assert np.array_equal(p.cpu().numpy(), ref_p)
    with pytest.raises(NotImplementedError):
        ops.get_max_preds(hm)
