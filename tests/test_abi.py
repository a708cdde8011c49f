"""CPU-side checks of the C-ABI boundary: the library builds/loads, exports every function that
include/nb_hip.h declares, and the host wrappers fail loudly instead of falling back."""
import ctypes as C
import os

import pytest
import torch

from neuralbody_amd import _lib, build, ops


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.lib()


def test_library_exports_every_header_symbol(lib):
    names = _lib.header_functions()
    assert len(names) >= 15
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and include/nb_hip.h drifted"
    for n in names:
        assert hasattr(lib, n), n
    assert lib.nb_abi_version() == _lib.ABI_VERSION == 20


def test_sizes_and_struct_layout(lib):
    # 8*44*256 + 256 + 2*(8*32*256 + 256) + 256 + 4 + 8*32*256 + 4*44*256 + 128 + 384 + 4
    # fp32 fragments + the f16f6 stream (4 waves x (132 one-KiB pieces + 4 KiB of scale dwords)) + its six-bit statistic
    assert lib.nb_mlp_pack_size() == 333320 + 4 * (132 * 1024 + 4096) // 4 + 8
    assert lib.nb_mlp_six_bit_stats_offset() == lib.nb_mlp_pack_size() - 8
    assert lib.nb_mlp_latent_bias_size() == 384
    assert C.sizeof(_lib.NbMlpParams) == 16 * 8
    assert lib.nb_scan_scratch_size(0) >= 256 and lib.nb_scan_scratch_size(1 << 20) >= 2 * 4 * (1 << 20)


def test_ctypes_structs_match_the_header_compiled_by_gcc(tmp_path):
    """sizeof / offsetof of nb_scene and nb_cull as a C compiler lays them out == the ctypes mirrors in _lib.py."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "nb_hip.h"\n'
        'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(nb_scene), offsetof(nb_scene, vol_dhw), '
        'offsetof(nb_scene, pose), offsetof(nb_scene, voxel_size), offsetof(nb_scene, out_sh), sizeof(nb_cull), '
        'offsetof(nb_cull, msk), offsetof(nb_cull, cam), offsetof(nb_cull, snap), sizeof(nb_mlp_params), offsetof(nb_scene, fold), '
        'sizeof(nb_fold), offsetof(nb_fold, grid), offsetof(nb_fold, row_base), offsetof(nb_fold, zero_row)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call([gcc, "-I", os.path.dirname(_lib.HEADER), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    S, K, F = _lib.NbScene, _lib.NbCull, _lib.NbFold
    want = [C.sizeof(S), S.vol_dhw.offset, S.pose.offset, S.voxel_size.offset, S.out_sh.offset, C.sizeof(K), K.msk.offset,
            K.cam.offset, K.snap.offset, C.sizeof(_lib.NbMlpParams), S.fold.offset, C.sizeof(F), F.grid.offset,
            F.row_base.offset, F.zero_row.offset]
    assert got == want, (got, want)


def test_ill_scratch_constants_match_the_header(tmp_path):
    """NB_ILL_* of include/nb_hip.h (the last-sample fix-up's scratch) == the numbers _lib.py sizes its buffer by."""
    import shutil
    import subprocess

    src = tmp_path / "ill.c"
    src.write_text('#include <stdio.h>\n#include "nb_hip.h"\nint main(void) { printf("%d %d %.9g %.9g\\n", (int)NB_ILL_SCRATCH_BYTES(1000), '
                   '(int)NB_ILL_SCRATCH_BYTES(0), (double)NB_ILL_SIGMA, (double)NB_ILL_T_MIN); return 0; }\n')
    exe = tmp_path / "ill"
    subprocess.check_call([shutil.which("gcc"), "-I", os.path.dirname(_lib.HEADER), str(src), "-o", str(exe)])
    got = subprocess.check_output([str(exe)]).split()
    assert int(got[0]) == _lib.ill_scratch_bytes(1000) and int(got[1]) == _lib.ill_scratch_bytes(0) == 64
    assert abs(float(got[2]) - _lib.ILL_SIGMA) < 1e-9 and abs(float(got[3]) - _lib.ILL_T_MIN) < 1e-12


def test_error_codes_without_touching_a_device(lib):
    # NULL scene -> NB_EINVAL and a message, no crash, no launch
    rc = lib.nb_march(None, None, None, None, None, None, None, 10, 64, None, None, None, 0, None, 0, None, None, None,
                      None, None, None, None, 0, 0, None)
    assert rc == -1
    assert b"nb_march" in lib.nb_last_error()
    rc = lib.nb_composite(None, None, None, 4, 0, 0, None, None, None, None, None, None)
    assert rc == -1
    rc = lib.nb_enc_conv(None, None, (C.c_int32 * 3)(1, 1, 1), None, None, 0, (C.c_int32 * 3)(1, 1, 1), 1, None, 16,
                         16, None, None, 0, None)
    assert rc == -1


def test_wrappers_refuse_cpu_tensors():
    with pytest.raises(_lib.NbError):
        ops.composite(torch.zeros(2, 64, 4), torch.zeros(2, 64), torch.zeros(2, 3))
    with pytest.raises(_lib.NbError):
        ops.enc_voxelize(torch.zeros(10, 3, dtype=torch.int32), [32, 32, 32])


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.NbError):
        _lib.lib()


def test_network_state_dict_matches_reference_keys():
    """120 entries with the reference's names/shapes (SURVEY.md §5 checkpoint row)."""
    from tests import synthetic as syn
    from neuralbody_amd.network import Network

    sd_ref = syn.make_weights(0, num_train_frame=5)
    net = Network(num_train_frame=5)
    sd = net.state_dict()
    assert len(sd) == 120
    assert set(sd) == set(sd_ref)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(sd_ref[k].shape), k
