"""No-GPU checks of the drop-in boundary and the host logic:
  * libwdhip.so loads and exports every symbol include/wd_hip.h declares;
  * the gfx950 code object exists and carries every kernel the managers ask for;
  * compute entry points fail loudly (never fall back) without a device;
  * DataManager / FunctionFeed / DataFeed bookkeeping (reference
    tests/warp_drive/pycuda_tests/test_data_manager.py:24-88 host-side parts)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge

    ge.build()
    from warp_drive_amd.managers import hip_driver as drv

    return drv


def test_header_symbols_are_exported(built):
    header = open(os.path.join(ROOT, "include", "wd_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(wd_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) >= 30
    lib = built.load_library()
    for sym in declared:
        assert hasattr(lib, sym), f"libwdhip.so does not export {sym}"
    assert sorted(built.C_ABI_SYMBOLS) == declared
    # the library must not be hard-linked against a particular HIP runtime
    needed = subprocess.run(["readelf", "-d", built.LIB_PATH], capture_output=True, text=True).stdout
    assert "amdhip64" not in needed


def test_header_is_plain_c_and_binds_from_c(built, tmp_path):
    """include/wd_hip.h must compile as C (cgo / JNI / N-API bindings) and the entry points must
    resolve and follow the error convention from a program that knows nothing about Python or torch"""
    exe = str(tmp_path / "c_abi_consumer")
    src = os.path.join(ROOT, "tests", "c", "c_abi_consumer.c")
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                    "-ldl"], check=True)
    out = subprocess.run([exe, built.LIB_PATH], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("version ")


def test_code_object_has_all_kernels(built):
    """every kernel the host asks for by name is in the manifest, in the code object the manifest names, built for
    gfx950; the main object holds the core services and the small envs, nothing test-only"""
    from warp_drive_amd import build as wd_build
    from warp_drive_amd.managers.function_manager import DEFAULT_FUNCTION_NAMES

    manifest = built.manifest()
    main = os.path.basename(built.HSACO_PATH)
    for name in list(DEFAULT_FUNCTION_NAMES) + ["HipTagGridWorldStep", "HipTagGridWorldTick", "HipTagGridWorldRollout",
                                                "HipClassicControlCartPoleEnvStep", "HipClassicControlCartPoleEnvTick",
                                                "testkernel"]:
        assert manifest.get(name) == main, f"{name} is not in the main code object"
    assert b"kIndexToActionArr" in open(built.HSACO_PATH, "rb").read()
    wanted = {"HipTagContinuousStep": "wd_kernels_tc.hsaco", "HipTagContinuousTick": "wd_kernels_tc.hsaco",
              "HipTagContinuousStep_K10": "wd_kernels_tc_k10.hsaco", "HipTagContinuousTick_K10": "wd_kernels_tc_k10.hsaco",
              "HipTagContinuousTickA_K10": "wd_kernels_tc_k10.hsaco",
              "HipTagContinuousTick_K10_N512": "wd_kernels_tc_k10.hsaco",
              "HipTagContinuousStep_K10_N1024": "wd_kernels_tc_k10.hsaco",
              "HipTagContinuousTick_K16_N1024": "wd_kernels_tc_k16.hsaco",
              "HipTagContinuousStep_K32_N512": "wd_kernels_tc_k32.hsaco",
              "HipTagContinuousTick_K10_N105A21": "wd_kernels_tc_k10_n105a21.hsaco",
              "HipTagContinuousTickA_K10_N105A21": "wd_kernels_tc_k10_n105a21.hsaco",
              "HipTagGridWorldRollout_N5": "wd_kernels_gw5.hsaco", "HipPolicyMlp_256x256_k3": "wd_kernels_mlp.hsaco", "HipWeightGradBx3_256x256": "wd_kernels_update.hsaco",
              "wd_test_math": "wd_kernels_test.hsaco", "wd_write_probe": "wd_kernels_test.hsaco"}
    for name, obj in wanted.items():
        assert manifest.get(name) == obj, (name, manifest.get(name))
    for obj in set(manifest.values()):
        assert obj in wd_build.UNITS
        blob = open(built.code_object_path(obj), "rb").read()
        assert b"gfx950" in blob
        assert all(n.encode() + b".kd" in blob for n, o in manifest.items() if o == obj)
    assert built.code_object_of("no_such_kernel") is None
    assert set(built.extra_code_objects()) == {built.code_object_path(o) for o in wd_build.UNITS} - {built.HSACO_PATH}


def test_no_silent_cpu_fallback(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from warp_drive_amd.env_wrapper import EnvWrapper
    from warp_drive_amd.envs.tag_gridworld import CUDATagGridWorld

    with pytest.raises(Exception):
        EnvWrapper(env_obj=CUDATagGridWorld(num_taggers=4), num_envs=2, env_backend="hip")
    lib = built.load_library()
    p = built._vp(0)
    import ctypes

    rc = lib.wd_malloc(16, ctypes.byref(p))
    assert rc != 0 and lib.wd_last_error()  # no runtime bound / no device: an error, not a fake success


class _HostOnlyDataManager:
    """CUDADataManager with device transfers replaced by host copies, to exercise the
    backend-independent bookkeeping without a GPU (test double)."""

    def __new__(cls, *a, **kw):
        from warp_drive_amd.managers.data_manager import CUDADataManager

        class _DM(CUDADataManager):
            def _to_device(self, name, name_on_device=None, torch_accessible=False):
                dst = name if name_on_device is None else name_on_device
                assert dst not in self._device_data_pointer
                self._device_data_pointer[dst] = self._host_data[name].copy()
                if torch_accessible:
                    import torch

                    self._device_data_via_torch[dst] = torch.from_numpy(self._device_data_pointer[dst])

            def pull_data_from_device(self, name):
                return self._host_data[name] if name in self._scalar_data_list else self._device_data_pointer[name]

        return _DM(*a, **kw)


def test_data_manager_bookkeeping():
    from warp_drive_amd.managers.function_manager import CUDAFunctionFeed
    from warp_drive_amd.utils.data_feed import DataFeed

    dm = _HostOnlyDataManager(num_agents=5, num_envs=2, episode_length=3)
    assert dm.meta_info("n_agents") == 5 and dm.meta_info("n_agents").dtype == np.int32
    for builtin in ("_done_", "_timestep_", "_log_mask_"):
        assert dm.is_data_on_device(builtin)
    assert dm.is_data_on_device_via_torch("_done_") and not dm.is_data_on_device_via_torch("_timestep_")
    assert dm.get_shape("_log_mask_") == (4,)
    f = DataFeed()
    f.add_data(name="X", data=np.array([[1, 2, 3, 4, 5], [0, 0, 0, 0, 0]]), save_copy_and_apply_at_reset=True,
               log_data_across_episode=True)
    f.add_data(name="Y", data=[[0.1, 0.2, 0.3, 0.4, 0.5], [0.0] * 5])
    f.add_data_list([("a", 100), ("b", 0.25, True), {"name": "c", "data": True}])
    dm.push_data_to_device(f)
    assert dm.get_dtype("X") == "int32" and dm.get_dtype("Y") == "float32"
    assert dm.reset_data_list == ["X"] and dm.log_data_list == ["X"]
    assert dm.get_shape("X_at_reset") == (2, 5) and dm.get_shape("X_for_log") == (4, 5)
    assert dm.device_data("a") == np.int32(100) and dm.device_data("b").dtype == np.float32
    assert dm.device_data("c") == 1 and dm.device_data("c").dtype == np.int32
    assert dm.scalar_data_list == ["a", "b", "c"]
    with pytest.raises(AssertionError):
        dm.push_data_to_device(f)
    dm.add_shared_constants({"kIndexToActionArr": [[0, 0], [1, 0], [-1, 0], [0, 1], [0, -1]], "k": 3})
    assert dm.shared_constant("kIndexToActionArr").dtype == np.int32 and dm.shared_constant("k") == 3
    with pytest.raises(AssertionError):
        dm.add_meta_info({"n_agents": 7})
    pool = DataFeed()
    pool.add_data(name="Z", data=np.zeros((2, 5), dtype=np.int32))
    pool.add_pool_for_reset(name="Z_reset_pool", data=np.ones((3, 5), dtype=np.int32), reset_target="Z")
    dm.push_data_to_device(pool)
    assert dm.reset_target_to_pool == {"Z": "Z_reset_pool"} and dm.get_reset_pool("Z") == "Z_reset_pool"
    feed = CUDAFunctionFeed(dm)
    args = feed(["X", "a", ("episode_length", "meta"), ("k", "shared"), ("Y", "device")])
    assert args[0] is dm.device_data("X") and args[1] == 100 and args[2] == 3 and args[3] == 3
    assert feed(["ignored"]) is args  # resolved once, then cached (function_manager.py:116-134)


def test_packed_geometry_fills_wavefronts(built):
    from warp_drive_amd.managers.function_manager import CUDAFunctionManager, HIPFunctionManager

    fm = CUDAFunctionManager(num_agents=105, num_envs=2000)
    assert fm.block == (105, 1, 1) and fm.grid == (2000, 1)  # the reference's default geometry
    geo = HIPFunctionManager.packed_geometry
    # the product geometry (envs/tag_continuous.py::_geometry: blocks of at most 256 threads):
    # 105 agents -> one replica per 128-thread block, one block per replica
    epb, block, grid = geo(fm, 105, 256)
    assert (epb, block, grid) == (1, (128, 1, 1), (2000, 1))
    # a 512-thread budget would pack 3 replicas into 5 wavefronts (315 of 320 lanes)
    assert geo(fm, 105, 512) == (3, (320, 1, 1), (667, 1))
    epb, block, grid = geo(fm, 5, 256)
    assert block[0] % 64 == 0 and epb == block[0] // 5 and epb * 5 / block[0] > 0.99
    assert grid[0] * epb >= 2000


def test_kernel_argument_packing(built):
    args = (built.DevicePtr(0x1000), np.int32(7), np.float32(0.5), built.DevicePtr(0x2000), True, 3, 2.0)
    raw = built._pack_args(args)
    assert len(raw) == 8 + 4 + 4 + 8 + 4 + 4 + 4
    assert np.frombuffer(raw[:8], np.uint64)[0] == 0x1000 and np.frombuffer(raw[8:12], np.int32)[0] == 7
    assert np.frombuffer(raw[16:24], np.uint64)[0] == 0x2000
    raw = built._pack_args((np.int32(1), built.DevicePtr(0x3000)))  # pointer after a 4-byte scalar is 8-aligned
    assert len(raw) == 16 and np.frombuffer(raw[8:], np.uint64)[0] == 0x3000


def test_config0_plumbing_cpu():
    """BASELINE config[0]: TagGridWorld 6x6, 4 taggers + 1 runner, num_envs=2, pure CPU step()."""
    from warp_drive_amd.env_wrapper import EnvWrapper
    from warp_drive_amd.envs.tag_gridworld import TagGridWorld

    envs = [EnvWrapper(env_obj=TagGridWorld(num_taggers=4, grid_length=6, episode_length=20, seed=3),
                       env_backend="cpu") for _ in range(2)]
    rng = np.random.RandomState(0)
    total_done = 0
    for e in envs:
        obs = e.reset()
        assert set(obs) == set(range(5)) and obs[0].shape == (21,)
    for _ in range(60):
        for e in envs:
            obs, rew, done, _ = e.step({a: int(rng.randint(5)) for a in range(5)})
            assert len(rew) == 5
            if done["__all__"]:
                total_done += 1
                e.reset()
    assert total_done >= 4


def test_kernels_do_not_spill_to_scratch(built, tmp_path):
    """Resource usage of every kernel of the build, read from the code objects' metadata notes: no VGPR spills and no
    scratch (private segment) in the objects the BASELINE configs run -- the main one, the BASELINE-shape TagContinuous
    entries, the TagGridWorld 5-agent rollouts (a reordering of two statements once cost the live-policy entry 5 483
    spilled registers and 13 x its tick time without failing a single parity test), the policy kernels; a few bytes at
    most anywhere else."""
    import re

    from warp_drive_amd import build as wd_build

    llvm = os.path.join(wd_build.ROCM, "lib", "llvm", "bin")
    strict = {"wd_kernels.hsaco", "wd_kernels_tc_k10_n105a21.hsaco", "wd_kernels_gw5.hsaco", "wd_kernels_mlp.hsaco",
              "wd_kernels_update.hsaco"}
    seen = 0
    for obj in wd_build.UNITS:
        elf = str(tmp_path / (obj + ".elf"))
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={built.code_object_path(obj)}",
                        f"--output={elf}"], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", elf], check=True, capture_output=True,
                               text=True).stdout
        names = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", notes)
        kernels = [(n, int(p), int(v)) for n, p, v in names if not n.startswith(("_", "hidden"))]
        kernels = [k for k in kernels if k[0] in built.manifest()]
        assert kernels, obj
        seen += len(kernels)
        for name, private, spills in kernels:
            limit = 0 if obj in strict else 64
            assert private <= limit and spills <= (0 if obj in strict else 8), (obj, name, private, spills)
    assert seen == len(built.manifest())


def test_code_objects_are_a_function_of_source_and_flags_only(built, tmp_path):
    """The PMC records under profiles/ are keyed by the sha256 of the code object that holds the kernel: a rebuild of
    unchanged sources -- in another checkout directory, through another output file name -- must give the same bytes
    (hipcc's default compilation-unit id hashes those paths; warp_drive_amd/build.py passes an explicit -cuid)."""
    import hashlib

    from warp_drive_amd import build as wd_build

    name = "wd_kernels_tc_k10_n105a21.hsaco"
    unit, flags = wd_build.UNITS[name]
    out = tmp_path / "another_name.bin"
    subprocess.run([wd_build._hipcc(), *wd_build.KERNEL_FLAGS, *wd_build._cuid(name), *flags,
                    os.path.join(wd_build.KDIR, unit), "-o", str(out)], check=True)
    sha = lambda path: hashlib.sha256(open(path, "rb").read()).hexdigest()
    assert sha(out) == sha(built.code_object_path(name))
    # ... and that is the object the counters of profiles/ were collected on
    import json

    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["E2000"]
    assert rec["hsaco_sha256"] == sha(built.code_object_path(name)), "profiles/pmc_traffic.json is stale: re-collect (scripts/collect_profiles.sh)"
    assert json.load(open(os.path.join(ROOT, "profiles", "pmc_mix.json")))["hsaco_sha256"] == rec["hsaco_sha256"]
