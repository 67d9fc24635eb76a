"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/snappy_hip.h declares, the reference-shaped drivers build against include/, and -- on a
machine without a GPU -- the product path fails loudly instead of falling back to a CPU path."""
import ctypes
import os
import re
import subprocess

import pytest

import build_util
from pkg_loader import load_package

ROOT = build_util.ROOT
REF = "/root/reference"


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "snappy_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b((?:sz_|rocc_)[a-z0-9_]+)\s*\(", hdr)))


def test_jni_shim_uses_only_the_declared_abi():
    """compression-accelerator_amd/drivers/jni/ (the north star's Scala/JVM driver): there is no JDK here, so the Scala
    side is never compiled; its contact with the library is pinned -- every sz_* function the shim calls and every
    SZ_* constant it names is declared in include/snappy_hip.h, every native method of SnappyHip.scala has its
    Java_snappyhip_SnappyHip_00024_<name> in the shim and the other way round.  (The C side IS compiled and run,
    against a stand-in jni.h: test_jni_shim_argument_checks_run below, test_gpu_boundary.py on the GPU.)"""
    jdir = os.path.join(ROOT, "compression-accelerator_amd", "drivers", "jni")
    c = open(os.path.join(jdir, "snappy_hip_jni.c")).read()
    c = re.sub(r"/\*.*?\*/", "", c, flags=re.S)
    hdr = open(os.path.join(ROOT, "include", "snappy_hip.h")).read()
    declared = set(_declared_symbols())
    used = set(re.findall(r"\b(sz_[a-z0-9_]+)\s*\(", c))
    assert used and used <= declared, used - declared
    consts = set(re.findall(r"\b(SZ_[A-Z0-9_]+)\b", c))
    defined = set(re.findall(r"#define\s+(SZ_[A-Z0-9_]+)\b", hdr))
    assert consts <= defined, consts - defined
    scala = open(os.path.join(jdir, "SnappyHip.scala")).read()
    natives = set(re.findall(r"@native\s+def\s+(\w+)", scala))
    shims = set(re.findall(r"Java_snappyhip_SnappyHip_00024_(\w+)\s*\(", c))
    assert natives == shims and len(natives) >= 4, natives ^ shims


def test_jni_shim_argument_checks_run(tmp_path):
    """The shim's own logic -- sign, bound and capacity checks, the varint probe, the long[] it returns -- executed:
    snappy_hip_jni.c built (-Wall -Werror) against tests/jni_stub/jni.h, driven by tests/jni_stub/jni_driver.c with faked
    direct buffers.  Every case here returns before a kernel would be launched, so it needs no GPU."""
    import json
    exe = build_util.cc_jni(str(tmp_path / "jni_driver"))
    out = json.loads(subprocess.check_output([exe, "args", str(tmp_path)]).decode())
    assert out == build_util.JNI_ARGUMENT_CASES, out


def test_library_exports_every_declared_symbol():
    P = load_package()
    L = ctypes.CDLL(P.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 16
    for s in declared:
        assert hasattr(L, s), "libsnappy_hip.so does not export %s" % s
    assert set(declared) == set(P.ABI_SYMBOLS)
    # ... and NOTHING else: no kernel launch stub, no C++ template instance, no __hip_cuid_* marker has default
    # visibility (-fvisibility=hidden + SZ_API on the declarations + csrc/exports.map)
    out = subprocess.check_output(["nm", "-D", "--defined-only", P.LIB_PATH]).decode()
    exported = set(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == set(declared), exported ^ set(declared)
    hdr = open(os.path.join(ROOT, "include", "snappy_hip.h")).read()
    assert hdr.count("SZ_API ") == len(declared) + 1, "every declaration carries SZ_API (+ its definition)"


def test_library_was_built_from_this_tree():
    """sz_version() names the sources the library was built from; the same hash, recomputed (tests/build_util.py),
    must come out of csrc/ + include/ as they are now -- a stale .so cannot pass for HEAD's.  `make source_id`
    agrees with both."""
    P = load_package()
    ver = P.lib().sz_version().decode()
    sid = build_util.source_id()
    assert re.search(r"src [0-9a-f]{16}\)$", ver), ver
    assert "src " + sid in ver, (ver, sid)
    made = subprocess.check_output(["make", "-s", "-C", build_util.CSRC, "source_id"]).decode().split()[-1]
    assert made == sid
    assert ("untuned toolchain" in ver) == ("HIP 7.2.26015" not in ver)


def test_no_torch_or_oracle_in_the_abi_library():
    P = load_package()
    out = subprocess.check_output(["readelf", "-d", P.LIB_PATH]).decode()
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    assert not any(("torch" in n) or ("c10" in n) or ("oracle" in n) or ("snappy.so" in n) for n in needed), needed
    # the product sources never mention the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "compression-accelerator_amd")):
        for f in files:
            if f.endswith((".hip", ".cpp", ".h", ".c", ".cc", ".py")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_" not in txt and "liboracle" not in txt, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p):
            assert "oracle_" not in open(p).read()


def test_sizing_and_varint_helpers_need_no_gpu():
    P = load_package()
    for n in (0, 1, 100, 65536, 10**6):
        assert P.max_compressed_length(n) == 32 + n + n // 6
    assert P.max_compressed_length(65536) == 76490 <= P.SLOT_STRIDE
    assert P.uncompressed_length(bytes([0x82, 0x01, 0xb0])) == 130
    assert P.uncompressed_length(bytes([0x80, 0x80, 0x04])) == 65536
    with pytest.raises(P.SnappyHipError):
        P.uncompressed_length(b"\xff\xff\xff\xff\xff")
    assert P.lib().sz_strerror(P.SZ_ENODEV) == b"no usable GPU"
    assert b"gfx950" in P.lib().sz_version()


def test_rocc_surface_argument_checks():
    P = load_package()
    rd = ctypes.c_uint64(7)
    L = P.lib()
    assert L.sz_rocc_cmd(2, P.ROCC_SET_LENGTH, 10, 0, ctypes.byref(rd)) == P.SZ_EINVAL  # wrong channel
    assert L.sz_rocc_cmd(3, 9, 0, 0, ctypes.byref(rd)) == P.SZ_EINVAL  # unknown funct
    assert L.sz_rocc_cmd(3, P.ROCC_SET_LENGTH, 130, 0, ctypes.byref(rd)) == P.SZ_OK and rd.value == 0


def test_fails_loudly_without_a_gpu(tmp_path):
    P = load_package()
    if P.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(P.SnappyHipError) as e:
        P.compress(b"hello hello hello hello")
    assert e.value.code == P.SZ_ENODEV
    exe = build_util.cc(os.path.join(build_util.DRIVERS, "rocc_compress.c"), str(tmp_path / "rocc_compress"))
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 3 and b"no usable GPU" in r.stderr


def test_own_drivers_build(tmp_path):
    build_util.cc(os.path.join(build_util.DRIVERS, "rocc_compress.c"), str(tmp_path / "a"))
    build_util.cxx(os.path.join(build_util.DRIVERS, "sw_benchmark.cc"), str(tmp_path / "b"))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_reference_drivers_build_unchanged(tmp_path):
    """tests/compression.c, tests/swcompression.cc and the FireSim workload's compress.cc compile
    and link UNCHANGED against include/ + libsnappy_hip.so."""
    build_util.cc(os.path.join(REF, "tests", "compression.c"), str(tmp_path / "c1"), force_shim=True)
    build_util.cxx(os.path.join(REF, "tests", "swcompression.cc"), str(tmp_path / "c2"))
    build_util.cxx(os.path.join(REF, "workloads", "snappy", "overlay", "root", "compress.cc"), str(tmp_path / "c3"))


def _isa_of(src, tmp_path):
    out = os.path.join(str(tmp_path), os.path.basename(src) + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-I" + os.path.join(build_util.ROOT, "include"), "-I" + os.path.dirname(src),
                           "-S", "--cuda-device-only", src, "-o", out])
    with open(out) as f:
        return f.read().splitlines()


def test_no_reloaded_spill_feeds_a_memory_instruction_scalar_base(tmp_path):
    """Suspect met in round 1: experimental builds of the multi-wave kernel that hung the grid had a
    kernel-argument pointer spilled to a VGPR lane and reloaded with v_readlane + s_nop 4 right in
    front of `global_atomic_add ... s[N:N+1]`; builds with the pointer still in SGPRs ran.  It was not
    the whole cause, but the pattern is kept out: the persistent kernel hands
    its queue counter to the atomic through VGPRs, and this test scans the ISA of every kernel for a
    v_readlane that writes the low half of an SGPR pair which a vector-memory instruction uses as its
    scalar base within the next dozen instructions."""
    csrc = os.path.join(build_util.ROOT, "compression-accelerator_amd", "csrc")
    vmem = re.compile(r"^\s*(global|flat|buffer|scratch)_\w+\s.*\bs\[(\d+):(\d+)\]")
    offenders = []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith(".hip"):
            continue
        lines = [l for l in _isa_of(os.path.join(csrc, name), tmp_path) if l.strip() and not l.lstrip().startswith(";")]
        for i, l in enumerate(lines):
            m = vmem.match(l)
            if not m:
                continue
            lo = "s%s," % m.group(2)
            pair = "s[%s:%s]," % (m.group(2), m.group(3))
            for j in range(i - 1, max(-1, i - 13), -1):  # the CLOSEST writer of the base's low half decides
                w = lines[j].split()
                if len(w) >= 2 and w[1] in (lo, pair):
                    if w[0] == "v_readlane_b32":
                        offenders.append("%s: %s  <-  %s" % (name, l.strip(), lines[j].strip()))
                    break  # (written by anything else in between: not a reloaded spill any more)
    assert not offenders, "\n".join(offenders)


def test_global_hash_table_lookups_bypass_the_vector_l1(tmp_path):
    """The multi-wave kernel's extra waves keep their hash tables in global memory and read back entries
    that other lanes of the same wave stored a moment ago.  The look-up must come from the L2 (where the
    wave's write-through stores arrive in issue order), not from a line the CU's vector L1 holds from
    before the store: every 16-bit global load of the compress kernels carries sc1 (agent scope)."""
    src = os.path.join(build_util.ROOT, "compression-accelerator_amd", "csrc", "snappy_compress.hip")
    lines = _isa_of(src, tmp_path)
    loads = [l.strip() for l in lines if re.match(r"^\s*global_load_ushort\s", l)]
    assert len(loads) >= 4, "no table look-ups found: did the table change its entry type?"
    bad = [l for l in loads if " sc1" not in l]
    assert not bad, "\n".join(bad)


def test_parse_state_of_the_compress_kernels_is_provably_wave_uniform(tmp_path):
    """The compress kernels keep their parse state (positions, masks, flags) in scalar registers only
    while the compiler can PROVE it wave-uniform; one join of a per-lane branch merged with a uniform
    one, or one loop left through a per-lane branch, and LLVM moves the whole state into vector
    registers (64-bit vector arithmetic for the masks, readfirstlane round trips).
    LLVM's own uniformity analysis is the judge here: no loop of the compress kernels other than the
    small per-lane copy loops may have a divergent exit, and none of the parse-state variables may be
    divergent inside the main loops."""
    opt = "/opt/rocm/lib/llvm/bin/opt"
    if not os.path.exists(opt):
        pytest.skip("no opt in this image")
    src = os.path.join(build_util.ROOT, "compression-accelerator_amd", "csrc", "snappy_compress.hip")
    ll = os.path.join(str(tmp_path), "sc.ll")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-discard-value-names",
                           "-I" + os.path.join(build_util.ROOT, "include"), "-I" + os.path.dirname(src),
                           "-S", "-emit-llvm", "--cuda-device-only", src, "-o", ll], stderr=subprocess.DEVNULL)
    r = subprocess.run([opt, "-passes=print<uniformity>", "-disable-output", ll], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = r.stderr
    state_names = r"(P|chain|sc|scan_base|e|t|S|jj|lim|P2|chain2|sc2|Hraw|Vacc|Hpacc|V|Hp|C|kcur|trips|c_in|sc_in)"
    state = re.compile(r"DIVERGENT:\s+%" + state_names + r"\.[0-9][\w.]* = ")
    # A loop with a divergent exit is one of the small per-lane copy loops exactly when it carries none of the
    # parse state: no phi of its entry blocks is one of the state variables.  (Read off the IR, not off the
    # compiler's block numbering: an allow-list of names like for.body53 breaks with the next unrelated edit.)
    state_phi = re.compile(r"^\s+%" + state_names + r"(\.[\w.]+)? = phi ")
    phis = {}  # function -> block label -> [phi lines]
    cur_fn, cur_bb = None, None
    for line in open(ll):
        if line.startswith("define "):
            cur_fn = line.split("@", 1)[1].split("(", 1)[0]
            phis[cur_fn] = {}
            cur_bb = "entry"
        elif line.startswith("}"):
            cur_fn = None
        elif cur_fn is not None:
            m = re.match(r"^([\w.$-]+):", line)
            if m:
                cur_bb = m.group(1)
            elif " = phi " in line:
                phis[cur_fn].setdefault(cur_bb, []).append(line)
    seen = 0
    for fn in text.split("UniformityInfo for function ")[1:]:
        name = fn.split("'")[1]
        if "compress_blocks" not in name and "compress_small" not in name:
            continue
        seen += 1
        assert name in phis, name
        cyc = fn.split("CYCLES WITH DIVERGENT EXIT:")[1].split("\n\n")[0] if "CYCLES WITH DIVERGENT EXIT:" in fn else ""
        bad = []
        for l in cyc.splitlines():
            m = re.search(r"entries\(([^)]*)\)", l)
            if not m:
                continue
            for bb in m.group(1).split():
                carried = [x.strip()[:80] for x in phis[name].get(bb, []) if state_phi.search(x)]
                if carried:
                    bad.append("%s carries %s" % (bb, carried[0]))
        assert not bad, "%s: loops left through a per-lane branch:\n%s" % (name, "\n".join(bad))
        div = [l.strip()[:120] for l in fn.splitlines() if state.search(l)]
        assert not div, "%s: parse state the compiler cannot prove uniform:\n%s" % (name, "\n".join(div[:10]))
    assert seen == 4  # single, multi-wave, team, small-blocks


def test_the_rccl_c_caller_builds_against_the_abi(tmp_path):
    """drivers/shard_compress.c (hipSetDevice, sz_shard_range, sz_compress_blocks_hip, ncclAllGather from librccl,
    sz_stream_offsets) compiles as plain C and links against include/ + libsnappy_hip.so + librccl; it runs in the
    GPU tests (tests/test_gpu_dist.py).  Without arguments it only prints its usage."""
    exe = build_util.cc_rccl(os.path.join(build_util.DRIVERS, "shard_compress.c"), str(tmp_path / "shard_compress"))
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 2 and b"usage" in r.stderr
