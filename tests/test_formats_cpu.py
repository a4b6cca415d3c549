"""CPU: the on-disk hand-offs (formats.py) round-trip with the layouts the reference reads/writes."""
import json

import numpy as np

from imagesequenceregistrationfor6dposeestimationlabeling_amd import formats, synth


def test_pose_files_roundtrip_with_failure_sentinel(tmp_path):
    rng = np.random.default_rng(0)
    R, t = synth.random_poses(rng, 5)
    R_list = [R[0], R[1], 1, R[3], R[4]]            # image 2 failed: pnp returned (1, 1, 1)
    t_list = [t[0], t[1], 1, t[3], t[4]]
    ok = formats.save_poses(R_list, t_list, 0, "tless", 1, base=tmp_path)
    assert ok.tolist() == [True, True, False, True, True]
    Rl, tl = formats.load_poses(0, "tless", 1, base=tmp_path)
    assert Rl.shape == (5, 3, 3) and tl.shape == (5, 3) and np.isnan(Rl[2]).all()
    np.testing.assert_array_equal(Rl[3], R[3])
    # the reference's own ragged object array is readable too
    d = tmp_path / formats.root_dir(1, "ruapc", 7)
    d.mkdir(parents=True)
    np.save(d / "7pred_R.npy", np.array([R[0], 1, R[2]], dtype=object), allow_pickle=True)
    np.save(d / "7pred_t.npy", np.array([t[0], 1, t[2]], dtype=object), allow_pickle=True)
    Rl, tl = formats.load_poses(1, "ruapc", 7, base=tmp_path)
    assert np.isnan(Rl[1]).all() and np.array_equal(tl[2], t[2])


def test_top_choices_and_vote_files(tmp_path):
    p = formats.write_top_choices(np.array([12, 3, 40]), 0, "ruapc", 1, base=tmp_path)
    assert p.read_text() == "12\n3\n40\n"
    assert formats.read_top_choices(0, "ruapc", 1, base=tmp_path) == [12, 3, 40]
    formats.save_vote(np.eye(3), [(0, 0), (1, 1)], 0, "ruapc", 1, base=tmp_path)
    assert np.load(tmp_path / "0_ruapc_obj_1" / "1error.npy").dtype == np.float64


def test_pred6d_json_producer_matches_the_reader_in_verfication(tmp_path):
    rng = np.random.default_rng(1)
    R, t = synth.random_poses(rng, 4)
    R[2] = np.nan
    formats.write_pred6d_json(R, t, [0, 1, 2, 10], tmp_path / "Tless" / "15poseEst_UH0" / "pred6d.json")
    ids, Rl, tl = formats.read_pred6d_json(tmp_path / "Tless" / "15poseEst_UH0" / "pred6d.json")
    assert ids == [0, 1, 10]
    raw = json.loads((tmp_path / "Tless" / "15poseEst_UH0" / "pred6d.json").read_text())
    assert np.allclose(np.array(raw["10"][0]["R"]).reshape(3, 3), R[3])       # verfication.py:76-79 indexing


def test_scene_json_and_crop_camera(tmp_path):
    gt = {"0": [{"cam_R_m2c": list(range(9)), "cam_t_m2c": [1, 2, 3], "obj_id": 1}],
          "10": [{"cam_R_m2c": [1, 0, 0, 0, 1, 0, 0, 0, 1], "cam_t_m2c": [0, 0, 700], "obj_id": 1}],
          "2": [{"cam_R_m2c": [0] * 9, "cam_t_m2c": [0, 0, 1], "obj_id": 1}]}
    (tmp_path / "scene_gt.json").write_text(json.dumps(gt))
    ids, R, t = formats.read_scene_gt(tmp_path / "scene_gt.json")
    assert ids == [0, 2, 10] and R.shape == (3, 3, 3) and t[2].tolist() == [0, 0, 700]
    K = np.array([[1075.65, 0, 360.0], [0, 1073.9, 270.0], [0, 0, 1]])
    cam = formats.crop_camera(K, (300, 200, 100, 80))
    # a point at the bbox centre must land on the centre of the 224 crop, then on the ::3 grid
    c = cam @ K_inv_point(K, 350.0, 240.0)
    assert abs(c[0] / c[2] - (112 + 0.5) / 3 + 0.5) < 1e-9 and abs(c[1] / c[2] - (112 + 0.5) / 3 + 0.5) < 1e-9


def K_inv_point(K, u, v):
    return np.linalg.inv(K) @ np.array([u, v, 1.0])


def test_crop_affine_of_a_one_pixel_box_raises_like_the_reference():
    """inference.py:203-215: an odd width / height is decremented, then size = 224 / max(w, h) / 1.2 — a 1 x 1 box (a single
    visible mask pixel) divides by zero there, and here (found by the front-end fuzz of tests/stress_gpu.py)."""
    import pytest
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import formats
    with pytest.raises(ZeroDivisionError):
        formats.crop_affine((17, 9, 1, 1))
    assert formats.crop_affine((17, 9, 2, 1)).shape == (2, 3)
