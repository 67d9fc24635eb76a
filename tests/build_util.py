"""Compiles the C/C++ drivers against include/ + libsnappy_hip.so into a temp dir."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
CSRC = os.path.join(ROOT, "compression-accelerator_amd", "csrc")
DRIVERS = os.path.join(ROOT, "compression-accelerator_amd", "drivers")
LINK = ["-L" + CSRC, "-lsnappy_hip", "-Wl,-rpath," + CSRC]


def cc(src, out, force_shim=False):
    cmd = ["gcc", "-std=gnu99", "-I" + INC]
    if force_shim:
        cmd += ["-include", os.path.join(INC, "rocc_host_shim.h")]
    subprocess.check_call(cmd + [src, "-o", out] + LINK, stderr=subprocess.DEVNULL)
    return out


def cxx(src, out):
    subprocess.check_call(["g++", "-std=c++11", "-I" + INC, src, "-o", out] + LINK, stderr=subprocess.DEVNULL)
    return out
