"""tools/verify_checkpoint.py — the one-command verifier for the rows that need a real checkpoint: its
structural half on CPU, its rendering half on the GPU against a reference-run fixture (the synthetic round trip: a checkpoint in
the reference's save_model format of the small scene's weights, the reference renderer's output for that scene)."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("verify_checkpoint", os.path.join(ROOT, "tools", "verify_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _checkpoint(tmp_path, sd_np, prefix="", epoch=7, name="latest.pth", mutate=None):
    sd = {prefix + k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}
    if mutate:
        mutate(sd)
    d = tmp_path / "trained_model"
    d.mkdir(exist_ok=True)
    torch.save({"net": sd, "optim": {}, "scheduler": {}, "recorder": {}, "epoch": epoch}, str(d / name))  # net_utils.py:319-329
    return str(d)


def _run(tool, argv, capsys):
    rc = tool.main(argv)
    out = capsys.readouterr()
    return rc, json.loads(out.out.strip().splitlines()[-1]), out.err


def test_structure_checks_need_no_gpu(tmp_path, capsys):
    tool = _tool()
    r, sd, body, batch, cam, _ = scenes.build("small")
    d = _checkpoint(tmp_path, sd, prefix="module.")  # saved from a DistributedDataParallel wrapper
    torch.save({"net": {}, "epoch": 1}, os.path.join(d, "3.pth"))
    rc, rep, err = _run(tool, [d, "--no-render"], capsys)  # a directory: latest.pth wins over 3.pth (load_network's rule)
    assert rc == 0 and rep["ok"] and rep["keys_ok"] and rep["layout_ok"] and rep["n_keys"] == 120 and rep["sparse_conv_weights"] == 17
    assert rep["checkpoint"].endswith("latest.pth") and rep["epoch"] == 7 and rep["num_train_frame"] == 7
    assert "structure ok" in err

    def broken(s):
        del s["xyzc_net.conv2.3.weight"]
        s["fc_1.weight"] = s["fc_1.weight"][:, :200]
        s["extra.bias"] = torch.zeros(3)

    d2 = _checkpoint(tmp_path / "b", sd, mutate=broken) if (tmp_path / "b").mkdir() is None else None
    rc, rep, err = _run(tool, [os.path.join(d2, "latest.pth"), "--no-render"], capsys)
    assert rc == 1 and not rep["ok"]
    assert rep["missing"] == ["xyzc_net.conv2.3.weight"] and rep["unexpected"] == ["extra.bias"]
    assert rep["shape_mismatch"] == {"fc_1.weight": [[256, 200, 1], [256, 256, 1]]}
    with pytest.raises(SystemExit, match="--batch and --reference"):
        tool.main([d])
    assert tool.find_checkpoint(d).endswith("latest.pth")
    os.remove(os.path.join(d, "latest.pth"))
    assert tool.find_checkpoint(d).endswith("3.pth")


@pytest.mark.gpu
@pytest.mark.parametrize("stored", ["as stored", "offsets mirrored (flip)"])
def test_synthetic_round_trip_renders_and_names_the_orientation(tmp_path, capsys, stored):
    """The small scene's weights as a reference-format checkpoint + the reference renderer's own output for that scene
    (tests/golden/scene_small.npz): VERIFIED as stored, every other orientation collapses; the same checkpoint written with its
    sparse kernels mirrored is diagnosed as exactly that."""
    tool = _tool()
    r, sd, body, batch, cam, _ = scenes.build("small")

    def mirror(s):
        for k in list(s):
            if k.startswith("xyzc_net.") and s[k].dim() == 5:
                s[k] = torch.flip(s[k], dims=(0, 1, 2)).contiguous()

    d = _checkpoint(tmp_path, sd, mutate=None if stored == "as stored" else mirror)
    np.savez(str(tmp_path / "batch.npz"), **batch)
    g = H.golden("small")
    np.savez(str(tmp_path / "ref.npz"), rgb_map=g["rgb_map"], weights=g["weights"], acc_map=g["acc_map"], depth_map=g["depth_map"],
             voxels_per_level=np.array([int(g["vol%d_nonzero_voxels" % l]) for l in range(4)]))
    rc, rep, err = _run(tool, [d, "--batch", str(tmp_path / "batch.npz"), "--reference", str(tmp_path / "ref.npz")], capsys)
    c = rep["candidates"]
    assert set(c) == set(tool.ORIENTATIONS)
    assert rep["best_orientation"] == stored and c[stored]["rgb_linf"] <= 5e-5
    others = [v["rgb_linf"] for k, v in c.items() if k != stored]
    assert min(others) > 1e-3, "a wrong kernel orientation must show as a collapse, not as rounding: %s" % c
    if stored == "as stored":
        assert rc == 0 and rep["ok"] and "VERIFIED" in err and rep["active_sets_match"]
    else:
        assert rc == 1 and "ORIENTATION" in err and "offsets mirrored (flip)" in err
