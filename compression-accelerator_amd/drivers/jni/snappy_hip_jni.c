/* snappy_hip_jni.c -- JNI shim of SnappyHip.scala: one C-ABI call per native method, no other logic.
 * Built for a JVM with the two commands in SnappyHip.scala (needs a JDK's jni.h: there is none in the build image).
 * In this repository the file is compiled and EXECUTED against a stand-in header, tests/jni_stub/jni.h, by a C program
 * that plays the JVM (tests/jni_stub/jni_driver.c: faked direct buffers; tests/test_abi.py runs the argument checks on
 * the CPU, tests/test_gpu_boundary.py::test_jni_shim_executes_on_the_gpu the round trips on the GPU); the Scala side
 * is not run.  tests/test_abi.py also checks that every sz_* symbol used here is declared in include/snappy_hip.h. */
#include <jni.h>
#include <stdint.h>

#include "snappy_hip.h"

static jlong rocc(int funct, void *s, jint len, void *d)
{
    uint64_t rd = 0;
    int rc = sz_rocc_cmd(SZ_ROCC_CHANNEL, SZ_ROCC_SET_LENGTH, (uint64_t)(uint32_t)len, 0, &rd);
    if (rc == SZ_OK)
        rc = sz_rocc_cmd(SZ_ROCC_CHANNEL, funct, (uint64_t)(uintptr_t)s, (uint64_t)(uintptr_t)d, &rd);
    return rc == SZ_OK ? (jlong)rd : (jlong)rc;
}

JNIEXPORT jlong JNICALL Java_snappyhip_SnappyHip_00024_maxCompressedLength(JNIEnv *e, jobject o, jlong n)
{
    (void)e; (void)o;
    return (jlong)sz_max_compressed_length((size_t)n);
}

JNIEXPORT jlong JNICALL Java_snappyhip_SnappyHip_00024_compress(JNIEnv *e, jobject o, jobject src, jint len, jobject dst)
{
    (void)o;
    void *s = (*e)->GetDirectBufferAddress(e, src), *d = (*e)->GetDirectBufferAddress(e, dst);
    if (!s || !d || len < 0 || (*e)->GetDirectBufferCapacity(e, src) < (jlong)len ||
        (*e)->GetDirectBufferCapacity(e, dst) < (jlong)sz_max_compressed_length((size_t)len))
        return (jlong)SZ_EINVAL;  /* (the RoCC command passes no bound for src or dst: checked here, where they are known) */
    return rocc(SZ_ROCC_COMPRESS, s, len, d);
}

JNIEXPORT jlong JNICALL Java_snappyhip_SnappyHip_00024_uncompress(JNIEnv *e, jobject o, jobject src, jint len, jobject dst)
{
    (void)o;
    void *s = (*e)->GetDirectBufferAddress(e, src), *d = (*e)->GetDirectBufferAddress(e, dst);
    size_t ulen = 0;
    if (!s || !d || len <= 0 || (*e)->GetDirectBufferCapacity(e, src) < (jlong)len)
        return (jlong)SZ_EINVAL;
    int rc = sz_get_uncompressed_length(s, (size_t)len, &ulen);
    if (rc != SZ_OK)
        return (jlong)rc;
    if ((*e)->GetDirectBufferCapacity(e, dst) < (jlong)ulen)
        return (jlong)SZ_ENOSPC;
    return rocc(SZ_ROCC_UNCOMPRESS, s, len, d);
}

JNIEXPORT jint JNICALL Java_snappyhip_SnappyHip_00024_compressBlocksDevice(JNIEnv *e, jobject o, jlong d_src, jint n_blocks,
                                                                          jlong d_dst, jlong slot_stride, jlong d_out_len,
                                                                          jlong stream)
{
    (void)e; (void)o;
    if (n_blocks < 0 || slot_stride <= 0)  /* (a negative jint would become a huge uint32 block count) */
        return (jint)SZ_EINVAL;
    return (jint)sz_compress_blocks_hip((const void *)(uintptr_t)d_src, SZ_BLOCK_SIZE, NULL, (uint32_t)n_blocks,
                                        (void *)(uintptr_t)d_dst, (size_t)slot_stride, (uint32_t *)(uintptr_t)d_out_len,
                                        SZ_FLAG_BLOCK_VARINT, (void *)(uintptr_t)stream);
}

JNIEXPORT jlongArray JNICALL Java_snappyhip_SnappyHip_00024_shardRange(JNIEnv *e, jobject o, jlong n_blocks, jint rank, jint world)
{
    (void)o;
    uint64_t b = 0, en = 0;
    if (sz_shard_range((uint64_t)n_blocks, (uint32_t)rank, (uint32_t)world, &b, &en) != SZ_OK)
        return NULL;
    jlong v[2] = {(jlong)b, (jlong)en};
    jlongArray a = (*e)->NewLongArray(e, 2);
    if (a)
        (*e)->SetLongArrayRegion(e, a, 0, 2, v);
    return a;
}
