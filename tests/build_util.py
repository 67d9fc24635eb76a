"""Compiles the C/C++ drivers against include/ + libsnappy_hip.so into a temp dir."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
CSRC = os.path.join(ROOT, "compression-accelerator_amd", "csrc")
DRIVERS = os.path.join(ROOT, "compression-accelerator_amd", "drivers")
LINK = ["-L" + CSRC, "-lsnappy_hip", "-Wl,-rpath," + CSRC]


def source_id():
    """The identity csrc/Makefile compiles into sz_version() (`src <id>`): sha256 over the library's sources --
    csrc/*.{hip,cpp,h}, exports.map, include/*.h, include/snappy/*.h -- concatenated in the order of their paths
    relative to csrc/ (make's $(sort)), first 16 hex digits."""
    import glob
    import hashlib
    rel = []
    for pat in ("*.hip", "*.cpp", "*.h", "../../include/*.h", "../../include/snappy/*.h", "exports.map"):
        rel += [os.path.relpath(p, CSRC) if not pat.startswith("..") else pat.rsplit("/", 1)[0] + "/" + os.path.basename(p)
                for p in glob.glob(os.path.join(CSRC, pat))]
    h = hashlib.sha256()
    for r in sorted(set(rel)):
        with open(os.path.join(CSRC, r), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def cc(src, out, force_shim=False):
    cmd = ["gcc", "-std=gnu99", "-I" + INC]
    if force_shim:
        cmd += ["-include", os.path.join(INC, "rocc_host_shim.h")]
    subprocess.check_call(cmd + [src, "-o", out] + LINK, stderr=subprocess.DEVNULL)
    return out


def cxx(src, out):
    subprocess.check_call(["g++", "-std=c++11", "-I" + INC, src, "-o", out] + LINK, stderr=subprocess.DEVNULL)
    return out


def cc_rccl(src, out):
    """A plain-C caller that also talks to the HIP runtime and to librccl directly (drivers/shard_compress.c)."""
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["gcc", "-std=gnu99", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-I" + INC,
                           src, "-o", out] + LINK + ["-L" + os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64",
                                                     "-Wl,-rpath," + os.path.join(rocm, "lib")], stderr=subprocess.DEVNULL)
    return out


def cc_jni(out):
    """drivers/jni/snappy_hip_jni.c (the C side of the Scala/JVM driver) built against tests/jni_stub/jni.h -- a test-only
    stand-in for the JDK's header -- together with tests/jni_stub/jni_driver.c, which plays the JVM."""
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    stub = os.path.join(ROOT, "tests", "jni_stub")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + stub, "-I" + INC,
                           "-I" + os.path.join(rocm, "include"), os.path.join(stub, "jni_driver.c"),
                           os.path.join(DRIVERS, "jni", "snappy_hip_jni.c"), "-o", out] + LINK +
                          ["-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    return out


JNI_ARGUMENT_CASES = {  # what the shim answers before anything reaches the GPU (SZ_EINVAL -22, SZ_ENOSPC -28, SZ_ECORRUPT -74)
    "max_65536": 76490, "compress_negative_len": -22, "compress_len_beyond_src": -22, "compress_short_dst": -22,
    "compress_heap_src": -22, "compress_heap_dst": -22, "compress_null_src": -22, "uncompress_zero_len": -22,
    "uncompress_negative_len": -22, "uncompress_len_beyond_src": -22, "uncompress_bad_varint": -74,
    "uncompress_short_dst": -28, "device_negative_blocks": -22, "device_bad_stride": -22, "shard_43_3_8": [18, 23],
    "shard_world_0_is_null": True, "shard_rank_beyond_world_is_null": True, "shard_alloc_failure_is_null": True,
}
