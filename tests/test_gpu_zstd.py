"""GPU parity of S3: agc_hip_zstd17_batch (one zstd frame per lane) must return, byte for byte, the frames
ZSTD_compressCCtx(level 17) of libzstd 1.4.9 writes for the same inputs."""
import numpy as np
import pytest

from tests import zstd_cases as ZC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("background", [0, 1])
def test_gpu_frames_equal_libzstd(hip_ctx, oracle, background):
    """both builds of the kernel: frequency tables in LDS (default) and in the workspace (agc_hip_zstd17_background: launches that
    run beside the steps of the next samples)"""
    if ZC.libzstd().ZSTD_versionNumber() != 10409:
        pytest.skip("parity is pinned against libzstd 1.4.9")
    inputs = ZC.corpus(oracle, 31337 + background, 400)
    assert hip_ctx.L.agc_hip_zstd17_background(hip_ctx.h, background) == 0
    try:
        got = hip_ctx.zstd17_batch(inputs)
    finally:
        hip_ctx.L.agc_hip_zstd17_background(hip_ctx.h, 0)
    bad = [(i, len(p)) for i, p in enumerate(inputs) if got[i] != ZC.ref_frame(p)]
    assert not bad, bad


@pytest.mark.parametrize("group", [0, 2, 3])
def test_gpu_group_kernels_equal_libzstd(hip_ctx, oracle, group):
    """the kernels with several lanes per frame (zstd/zs_opt_grp.h; 3 = the default, 2, and 0 = the one-lane kernel only) on the
    inputs they take (<= 16 KiB) mixed with ones they leave to the one-lane kernel"""
    import os
    if ZC.libzstd().ZSTD_versionNumber() != 10409:
        pytest.skip("parity is pinned against libzstd 1.4.9")
    rng = np.random.default_rng(40 + group)
    inputs = ZC.corpus(oracle, 555 + group, 300, max_len=16384) + ZC.corpus(oracle, 9, 40)
    inputs += [ZC.delta_pack(oracle, rng, n, 60000, 1e-3)[:16384] for n in (5, 25, 25, 30)]
    inputs += [b"A" * 9000, b"AB" * 4000, b"ABC" * 3000, (b"0,1234.C" * 40 + b"!!!!!!!!!!!!") * 30]
    os.environ["AGC_HIP_ZSTD_GROUP"] = str(group)
    try:
        got = hip_ctx.zstd17_batch(inputs)
    finally:
        del os.environ["AGC_HIP_ZSTD_GROUP"]
    bad = [(i, len(p)) for i, p in enumerate(inputs) if got[i] != ZC.ref_frame(p)]
    assert not bad, bad


def test_gpu_frames_from_inputs_resident_in_hbm(hip_ctx, oracle):
    """agc_hip_zstd17_batch_dev: the packs are in HBM already (the multi-GPU Close: another rank sent them); a run that does not
    start at the buffer's first byte"""
    import torch
    rng = np.random.default_rng(12)
    inputs = [ZC.delta_pack(oracle, rng, n, 60000, 1e-3) for n in (3, 25, 30, 60)] + ZC.corpus(oracle, 77, 40, max_len=40000)
    src = np.frombuffer(b"".join(inputs), np.uint8)
    off = np.zeros(len(inputs) + 1, np.uint64)
    off[1:] = np.cumsum([len(p) for p in inputs])
    d = torch.from_numpy(src.copy()).cuda()
    torch.cuda.synchronize()
    for a, b in ((0, len(inputs)), (5, 31)):
        frames, foff = hip_ctx.zstd17_batch_raw_dev(d.data_ptr(), off[a:b + 1])
        for i in range(a, b):
            got = frames[int(foff[i - a]):int(foff[i - a + 1])].tobytes()
            assert got == ZC.ref_frame(inputs[i]), (a, b, i)


def test_gpu_many_equal_size_packs(hip_ctx, oracle):
    """the shape Close() produces: thousands of packs of similar size in one call (several workspace-arena rounds when the
    arena is small)"""
    import os
    rng = np.random.default_rng(5)
    base = [ZC.delta_pack(oracle, rng, 20, 60000, 1e-3) for _ in range(40)]
    inputs = [base[i % 40][: len(base[i % 40]) - (i % 7)] for i in range(3000)]
    os.environ["AGC_HIP_ZSTD_ARENA_MB"] = "600"  # forces several rounds
    try:
        got = hip_ctx.zstd17_batch(inputs)
    finally:
        del os.environ["AGC_HIP_ZSTD_ARENA_MB"]
    want = {}
    for i, p in enumerate(inputs):
        if p not in want:
            want[p] = ZC.ref_frame(p)
        assert got[i] == want[p], i


@pytest.mark.gpu
def test_gpu_frames_of_levels_13_and_19_equal_libzstd(hip_ctx):
    """agc_hip_zstd_batch: tuple-packed references at level 13 (lane-group kernel up to 16 KiB, one-lane kernel above: btopt,
    minMatch 4), repetitive references at level 19 (btultra2 at every size), delta packs at level 17 -- mixed in one call"""
    from tests.test_zstd_frames import reference_like_inputs
    rng = np.random.default_rng(1319)
    items = reference_like_inputs(rng)
    items += [(17, bytes(rng.integers(65, 69, 9000, dtype=np.uint8))), (13, b""), (19, b"A" * 3), (13, b"ACGT" * 2000)]
    got = hip_ctx.zstd_batch([d for _, d in items], [lv for lv, _ in items])
    bad = [(lv, len(d)) for (lv, d), g in zip(items, got) if g != ZC.ref_frame(d, lv)]
    assert not bad, bad
