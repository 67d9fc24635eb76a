/* tests/jni_stub/jni_driver.c -- TEST INFRASTRUCTURE: stands in for a JVM in front of drivers/jni/snappy_hip_jni.c
 * (see jni.h beside this file).  Fakes direct ByteBuffers / long[] and calls the shim's entry points in the call shape
 * of the reference's driver (tests/compression.c:20-39: SET_LENGTH, COMPRESS -> rd, SET_LENGTH, UNCOMPRESS -> rd).
 *
 *   jni_driver args <out-dir>          argument checks only -- no GPU needed (every case returns before a kernel)
 *   jni_driver gpu  <out-dir> <file>.. the round trips on the GPU: per input file <out-dir>/<k>.snappy = what
 *                                      compress() wrote, + the device entry point on the first file's blocks
 * Prints one JSON object; the Python test compares the bytes with the oracle (this program knows no oracle). */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "snappy_hip.h"

/* ---- the stand-in "JVM": direct buffers and long arrays are small structs */
typedef struct { void *addr; jlong cap; } FakeBuffer;   /* addr == NULL: a heap (non-direct) ByteBuffer */
typedef struct { jsize len; jlong v[8]; } FakeLongArray;
static int g_fail_alloc;

static void *get_addr(JNIEnv *e, jobject b) { (void)e; return b ? ((FakeBuffer *)b)->addr : NULL; }
static jlong get_cap(JNIEnv *e, jobject b) { (void)e; return b && ((FakeBuffer *)b)->addr ? ((FakeBuffer *)b)->cap : -1; }
static jlongArray new_longs(JNIEnv *e, jsize n)
{
    (void)e;
    if (g_fail_alloc || n > 8)
        return NULL;
    FakeLongArray *a = (FakeLongArray *)calloc(1, sizeof *a);
    a->len = n;
    return a;
}
static void set_longs(JNIEnv *e, jlongArray a, jsize start, jsize n, const jlong *v)
{
    (void)e;
    memcpy(((FakeLongArray *)a)->v + start, v, (size_t)n * sizeof(jlong));
}
static const struct JNINativeInterface_ g_table = {get_addr, get_cap, new_longs, set_longs};
static JNIEnv g_env = &g_table;

/* ---- the shim's entry points (SnappyHip.scala's @native methods) */
jlong Java_snappyhip_SnappyHip_00024_maxCompressedLength(JNIEnv *, jobject, jlong);
jlong Java_snappyhip_SnappyHip_00024_compress(JNIEnv *, jobject, jobject, jint, jobject);
jlong Java_snappyhip_SnappyHip_00024_uncompress(JNIEnv *, jobject, jobject, jint, jobject);
jint Java_snappyhip_SnappyHip_00024_compressBlocksDevice(JNIEnv *, jobject, jlong, jint, jlong, jlong, jlong, jlong);
jlongArray Java_snappyhip_SnappyHip_00024_shardRange(JNIEnv *, jobject, jlong, jint, jint);

#define COMPRESS(s, n, d) Java_snappyhip_SnappyHip_00024_compress(&g_env, NULL, (s), (n), (d))
#define UNCOMPRESS(s, n, d) Java_snappyhip_SnappyHip_00024_uncompress(&g_env, NULL, (s), (n), (d))

static void *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return NULL;
    fseek(f, 0, SEEK_END);
    *n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc(*n + 1);
    if (fread(p, 1, *n, f) != *n)
        return NULL;
    fclose(f);
    return p;
}

static int spill(const char *dir, const char *name, const void *p, size_t n)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(p, 1, n, f) != n)
        return -1;
    return fclose(f);
}

static int argument_cases(void)
{
    static uint8_t in[4096], out[8192];
    FakeBuffer src = {in, 4096}, dst = {out, 8192}, heap = {NULL, 4096}, small = {out, 100};
    memset(in, 'a', sizeof in);
    printf("\"max_65536\": %lld, ", (long long)Java_snappyhip_SnappyHip_00024_maxCompressedLength(&g_env, NULL, 65536));
    printf("\"compress_negative_len\": %lld, ", (long long)COMPRESS(&src, -1, &dst));
    printf("\"compress_len_beyond_src\": %lld, ", (long long)COMPRESS(&src, 4097, &dst));
    printf("\"compress_short_dst\": %lld, ", (long long)COMPRESS(&src, 4096, &small));  /* < MaxCompressedLength(4096) */
    printf("\"compress_heap_src\": %lld, ", (long long)COMPRESS(&heap, 16, &dst));
    printf("\"compress_heap_dst\": %lld, ", (long long)COMPRESS(&src, 16, &heap));
    printf("\"compress_null_src\": %lld, ", (long long)COMPRESS(NULL, 16, &dst));
    printf("\"uncompress_zero_len\": %lld, ", (long long)UNCOMPRESS(&src, 0, &dst));
    printf("\"uncompress_negative_len\": %lld, ", (long long)UNCOMPRESS(&src, -5, &dst));
    printf("\"uncompress_len_beyond_src\": %lld, ", (long long)UNCOMPRESS(&src, 5000, &dst));
    static uint8_t bad_varint[8] = {0xff, 0xff, 0xff, 0xff, 0xff, 0, 0, 0};
    FakeBuffer bv = {bad_varint, 8};
    printf("\"uncompress_bad_varint\": %lld, ", (long long)UNCOMPRESS(&bv, 8, &dst));
    static uint8_t big[4] = {0x80, 0x80, 0x04, 0x00};  /* announces 65 536 bytes */
    FakeBuffer bg = {big, 4};
    printf("\"uncompress_short_dst\": %lld, ", (long long)UNCOMPRESS(&bg, 4, &small));
    printf("\"device_negative_blocks\": %d, ", (int)Java_snappyhip_SnappyHip_00024_compressBlocksDevice(&g_env, NULL, 4096, -3, 4096, SZ_SLOT_STRIDE, 4096, 0));
    printf("\"device_bad_stride\": %d, ", (int)Java_snappyhip_SnappyHip_00024_compressBlocksDevice(&g_env, NULL, 4096, 3, 4096, 0, 4096, 0));
    FakeLongArray *r = (FakeLongArray *)Java_snappyhip_SnappyHip_00024_shardRange(&g_env, NULL, 43, 3, 8);
    printf("\"shard_43_3_8\": [%lld, %lld], ", r ? (long long)r->v[0] : -1, r ? (long long)r->v[1] : -1);
    printf("\"shard_world_0_is_null\": %s, ", Java_snappyhip_SnappyHip_00024_shardRange(&g_env, NULL, 43, 0, 0) ? "false" : "true");
    printf("\"shard_rank_beyond_world_is_null\": %s, ", Java_snappyhip_SnappyHip_00024_shardRange(&g_env, NULL, 43, 8, 8) ? "false" : "true");
    g_fail_alloc = 1;  /* a JVM out of memory: NewLongArray returns null, the shim passes it on */
    printf("\"shard_alloc_failure_is_null\": %s", Java_snappyhip_SnappyHip_00024_shardRange(&g_env, NULL, 43, 3, 8) ? "false" : "true");
    g_fail_alloc = 0;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s args|gpu <out-dir> [files...]\n", argv[0]);
        return 2;
    }
    printf("{");
    argument_cases();
    if (strcmp(argv[1], "gpu") != 0) {
        printf("}\n");
        return 0;
    }
    printf(", \"files\": [");
    for (int k = 3; k < argc; k++) {
        size_t n = 0;
        uint8_t *data = (uint8_t *)slurp(argv[k], &n);
        if (!data)
            return 2;
        const size_t cap = sz_max_compressed_length(n);
        uint8_t *comp = (uint8_t *)malloc(cap), *back = (uint8_t *)malloc(n + 1);
        FakeBuffer src = {data, (jlong)n}, dst = {comp, (jlong)cap}, rt = {back, (jlong)n};
        const jlong clen = COMPRESS(&src, (jint)n, &dst);           /* SET_LENGTH + COMPRESS -> rd */
        char name[64];
        snprintf(name, sizeof name, "%d.snappy", k - 3);
        if (clen > 0 && spill(argv[2], name, comp, (size_t)clen) != 0)
            return 2;
        FakeBuffer csrc = {comp, clen > 0 ? clen : 0};
        const jlong urc = clen > 0 ? UNCOMPRESS(&csrc, (jint)clen, &rt) : -1;  /* rd == 0: success */
        const int same = urc == 0 && memcmp(back, data, n) == 0;
        /* a damaged copy of the same stream: an offset that reaches before the start of the output */
        jlong bad_rc = -1;
        if (clen > 8) {
            uint8_t *bad = (uint8_t *)malloc((size_t)clen);
            memcpy(bad, comp, (size_t)clen);
            size_t body = 0;
            while (bad[body] & 0x80)
                body++;
            body++;
            bad[body] = 0x02 | (3 << 2);  /* copy-2, length 4 ... */
            bad[body + 1] = 0xff;         /* ... offset 65 535 with nothing written yet */
            bad[body + 2] = 0xff;
            FakeBuffer bsrc = {bad, clen};
            bad_rc = UNCOMPRESS(&bsrc, (jint)clen, &rt);
            free(bad);
        }
        /* the decoded size does not fit dst */
        FakeBuffer tight = {back, (jlong)n - 1};
        const jlong tight_rc = clen > 0 && n > 0 ? UNCOMPRESS(&csrc, (jint)clen, &tight) : 0;
        printf("%s{\"n\": %zu, \"compressed\": %lld, \"uncompress_rd\": %lld, \"round_trip\": %s, \"damaged_rd\": %lld, \"tight_dst\": %lld}",
               k > 3 ? ", " : "", n, (long long)clen, (long long)urc, same ? "true" : "false", (long long)bad_rc, (long long)tight_rc);
        free(comp);
        free(back);
        free(data);
    }
    printf("]");
    /* the device entry point: the first file's 64 KiB blocks, resident in HBM, slots + lengths copied back */
    if (argc > 3) {
        size_t n = 0;
        uint8_t *data = (uint8_t *)slurp(argv[3], &n);
        const uint32_t nb = (uint32_t)(n / SZ_BLOCK_SIZE);
        if (nb) {
            void *d_src = NULL, *d_dst = NULL, *d_len = NULL;
            if (hipMalloc(&d_src, (size_t)nb * SZ_BLOCK_SIZE) != hipSuccess || hipMalloc(&d_dst, (size_t)nb * SZ_SLOT_STRIDE) != hipSuccess ||
                hipMalloc(&d_len, (size_t)nb * 4) != hipSuccess ||
                hipMemcpy(d_src, data, (size_t)nb * SZ_BLOCK_SIZE, hipMemcpyHostToDevice) != hipSuccess)
                return 3;
            const jint rc = Java_snappyhip_SnappyHip_00024_compressBlocksDevice(&g_env, NULL, (jlong)(uintptr_t)d_src, (jint)nb, (jlong)(uintptr_t)d_dst,
                                                                                 SZ_SLOT_STRIDE, (jlong)(uintptr_t)d_len, 0);
            uint8_t *slots = (uint8_t *)malloc((size_t)nb * SZ_SLOT_STRIDE);
            uint32_t *lens = (uint32_t *)malloc((size_t)nb * 4);
            if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(slots, d_dst, (size_t)nb * SZ_SLOT_STRIDE, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(lens, d_len, (size_t)nb * 4, hipMemcpyDeviceToHost) != hipSuccess)
                return 3;
            /* a slot stride below the kernel's bound: the C ABI's own error comes back through the shim */
            const jint rc_tight = Java_snappyhip_SnappyHip_00024_compressBlocksDevice(&g_env, NULL, (jlong)(uintptr_t)d_src, (jint)nb, (jlong)(uintptr_t)d_dst,
                                                                                       70000, (jlong)(uintptr_t)d_len, 0);
            printf(", \"device\": {\"rc\": %d, \"rc_tight_stride\": %d, \"blocks\": %u, \"lens\": [", (int)rc, (int)rc_tight, nb);
            for (uint32_t b = 0; b < nb; b++) {
                printf("%s%u", b ? ", " : "", lens[b]);
                char name[64];
                snprintf(name, sizeof name, "dev_%u.snappy", b);
                if (lens[b] <= SZ_SLOT_STRIDE && spill(argv[2], name, slots + (size_t)b * SZ_SLOT_STRIDE, lens[b]) != 0)
                    return 2;
            }
            printf("]}");
            (void)hipFree(d_src);
            (void)hipFree(d_dst);
            (void)hipFree(d_len);
        }
    }
    printf("}\n");
    sz_release_thread_resources();
    return 0;
}
