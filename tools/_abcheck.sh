mkdir -p gpurun_out/r5_10
python tools/abmulti.py ab/libsnappy_hip_cls.so,ab/libsnappy_hip_spec3.so text:2048,c5:2048,mix:3072 2 --check 2>&1 | tee gpurun_out/r5_10/check.log
SZ_KERNEL=dual python tools/abmulti.py ab/libsnappy_hip_cls.so,ab/libsnappy_hip_spec3.so text:2048,mix:3072 2 --check 2>&1 | tee -a gpurun_out/r5_10/check.log
python tools/abx.py ab/libsnappy_hip_cls.so,ab/libsnappy_hip_spec3.so,ab/libsnappy_hip_cls.so@dual,ab/libsnappy_hip_spec3.so@dual text:16384,c5:16384,mix:16386 7 2>&1 | tee gpurun_out/r5_10/ab.log
