/*
 * shard_compress.c -- one rank of the sharded path in plain C: the sequence INTEGRATION.md section 6 prints, compiled
 * and run (north star: "a C-ABI HIP launcher ... RCCL only for the final length gather").
 *
 *   hipSetDevice(local_rank) -> sz_shard_range -> sz_compress_blocks_hip on this rank's blocks (no data-path
 *   collective) -> ncclAllGather of the uint32 lengths, from librccl directly -> sz_stream_offsets.
 *
 * Build:  hipcc -x c shard_compress.c -Iinclude -Lcsrc -lsnappy_hip -lrccl        (tests/test_gpu_dist.py does)
 * Usage:  shard_compress <input-file> <world> <rank> <id-file> [device]
 *   The job = the file cut into 64 KiB blocks (the last one ragged).  Rank 0 writes the RCCL unique id to <id-file>
 *   (its name + ".tmp", renamed when complete), the other ranks wait for it: no MPI, no launcher.  `device` defaults
 *   to `rank`.  Every rank prints ONE line of JSON: the gathered lengths, the layout's total, and what it checked
 *   (its own slice of the gathered lengths is what its kernel wrote; the layout ends at the sum).  Exit status 0 iff
 *   every step succeeded.  The caller compares the lengths with the oracle's (this program knows no oracle).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "snappy_hip.h"

/* A rank that fails after the communicator exists ABORTS it on its way out (ncclCommAbort): its peers then get an
 * error from ncclCommInitRank / ncclAllGather instead of waiting for this rank for ever. */
static ncclComm_t g_comm;
static int g_comm_up;
static int leave(int code)
{
    if (g_comm_up) {
        g_comm_up = 0;
        (void)ncclCommAbort(g_comm);
    }
    return code;
}
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return leave(3); } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return leave(4); } } while (0)
#define CHECK_SZ(x) do { int c_ = (x); if (c_ != SZ_OK) { fprintf(stderr, "%s: %s\n", #x, sz_strerror(c_)); return leave(5); } } while (0)

/* What a rank HAD, published beside its lengths (bench.py's verify_gather does the same): rank, number of blocks, sum
 * and a position-weighted checksum of its lengths.  Every rank compares each peer's tag with the slice it received. */
static void tag_of(const uint32_t *lens, uint64_t n, uint32_t rank, uint64_t tag[4])
{
    uint64_t sum = 0, chk = 0;
    for (uint64_t k = 0; k < n; k++) {
        sum += lens[k];
        chk = (chk + (uint64_t)lens[k] * (k + 1)) % 2305843009213693951ull;
    }
    tag[0] = rank;
    tag[1] = n;
    tag[2] = sum;
    tag[3] = chk;
}

static int share_id(ncclUniqueId *id, const char *path, int rank)
{
    if (rank == 0) {
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", path);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, sizeof *id, 1, f) != 1)
            return -1;
        fclose(f);
        return rename(tmp, path);
    }
    for (int tries = 0; tries < 6000; tries++) {  /* up to a minute */
        FILE *f = fopen(path, "rb");
        if (f) {
            const size_t got = fread(id, sizeof *id, 1, f);
            fclose(f);
            if (got == 1)
                return 0;
        }
        usleep(10000);
    }
    return -1;
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s <input-file> <world> <rank> <id-file> [device]\n", argv[0]);
        return 2;
    }
    const uint32_t world = (uint32_t)atoi(argv[2]), rank = (uint32_t)atoi(argv[3]);
    const int device = argc > 5 ? atoi(argv[5]) : (int)rank;
    FILE *f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 2;
    }
    fseek(f, 0, SEEK_END);
    const uint64_t n_bytes = (uint64_t)ftell(f);
    const uint64_t n_blocks_total = (n_bytes + SZ_BLOCK_SIZE - 1) / SZ_BLOCK_SIZE;

    /* this rank's blocks [b, e) of the job; the entry points follow the calling thread's current device */
    uint64_t b = 0, e = 0;
    CHECK_SZ(sz_shard_range(n_blocks_total, rank, world, &b, &e));
    const uint32_t mine = (uint32_t)(e - b);
    const uint64_t per = (n_blocks_total + world - 1) / world;  /* ncclAllGather wants equal counts: pad to the longest shard */
    CHECK_HIP(hipSetDevice(device));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    /* input: this rank's bytes only (the last block of the job may be ragged: explicit lengths) */
    const uint64_t first = b * SZ_BLOCK_SIZE, last = e * SZ_BLOCK_SIZE < n_bytes ? e * SZ_BLOCK_SIZE : n_bytes;
    uint8_t *h_src = (uint8_t *)calloc((size_t)mine + 1, SZ_BLOCK_SIZE);
    uint32_t *h_in_len = (uint32_t *)calloc((size_t)mine + 1, sizeof(uint32_t));
    fseek(f, (long)first, SEEK_SET);
    if (last > first && fread(h_src, 1, (size_t)(last - first), f) != (size_t)(last - first))
        return 2;
    fclose(f);
    for (uint32_t k = 0; k < mine; k++) {
        const uint64_t left = (last - first) - (uint64_t)k * SZ_BLOCK_SIZE;
        h_in_len[k] = (uint32_t)(left < SZ_BLOCK_SIZE ? left : SZ_BLOCK_SIZE);
    }
    uint8_t *d_src, *d_slots;
    uint32_t *d_in_len, *d_len, *d_all;
    uint64_t *d_tag, *d_tags;
    CHECK_HIP(hipMalloc((void **)&d_src, ((size_t)mine + 1) * SZ_BLOCK_SIZE));
    CHECK_HIP(hipMalloc((void **)&d_slots, ((size_t)mine + 1) * SZ_SLOT_STRIDE));
    CHECK_HIP(hipMalloc((void **)&d_in_len, ((size_t)mine + 1) * 4));
    CHECK_HIP(hipMalloc((void **)&d_len, (size_t)per * 4));
    CHECK_HIP(hipMalloc((void **)&d_all, (size_t)per * world * 4));
    CHECK_HIP(hipMalloc((void **)&d_tag, 4 * 8));
    CHECK_HIP(hipMalloc((void **)&d_tags, (size_t)world * 4 * 8));
    CHECK_HIP(hipMemsetAsync(d_len, 0, (size_t)per * 4, stream));  /* (the padding of a short shard gathers as 0) */
    CHECK_HIP(hipMemsetAsync(d_all, 0xff, (size_t)per * world * 4, stream));  /* what did not arrive reads as a marker, never as a length */
    CHECK_HIP(hipMemsetAsync(d_tags, 0xff, (size_t)world * 4 * 8, stream));
    CHECK_HIP(hipMemcpyAsync(d_src, h_src, (size_t)mine * SZ_BLOCK_SIZE, hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemcpyAsync(d_in_len, h_in_len, (size_t)mine * 4, hipMemcpyHostToDevice, stream));

    /* the one communicator: the id travels through a file */
    ncclUniqueId id;
    memset(&id, 0, sizeof id);
    if (rank == 0)
        CHECK_NCCL(ncclGetUniqueId(&id));
    if (share_id(&id, argv[4], (int)rank) != 0) {
        fprintf(stderr, "rank %u: no RCCL unique id through %s\n", rank, argv[4]);
        return 4;
    }
    CHECK_NCCL(ncclCommInitRank(&g_comm, (int)world, id, (int)rank));
    g_comm_up = 1;
    const ncclComm_t comm = g_comm;

    /* the hot path: bodies + lengths of this rank's blocks, no data-path collective */
    CHECK_SZ(sz_compress_blocks_hip(d_src, SZ_BLOCK_SIZE, d_in_len, mine, d_slots, SZ_SLOT_STRIDE, d_len, 0, stream));
    /* the one exchange: the lengths */
    CHECK_NCCL(ncclAllGather(d_len, d_all, (size_t)per, ncclUint32, comm, stream));
    uint32_t *h_pad = (uint32_t *)malloc((size_t)per * world * 4), *h_all = (uint32_t *)malloc(((size_t)n_blocks_total + 1) * 4);
    uint32_t *h_mine = (uint32_t *)malloc(((size_t)mine + 1) * 4);
    CHECK_HIP(hipMemcpyAsync(h_pad, d_all, (size_t)per * world * 4, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(h_mine, d_len, (size_t)mine * 4, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    /* outside the path: every rank publishes the tag of what it had (a second, 32-byte all-gather) */
    uint64_t my_tag[4], *h_tags = (uint64_t *)malloc((size_t)world * 4 * 8);
    tag_of(h_mine, mine, rank, my_tag);
    CHECK_HIP(hipMemcpyAsync(d_tag, my_tag, sizeof my_tag, hipMemcpyHostToDevice, stream));
    CHECK_NCCL(ncclAllGather(d_tag, d_tags, 4, ncclUint64, comm, stream));
    CHECK_HIP(hipMemcpyAsync(h_tags, d_tags, (size_t)world * 4 * 8, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    uint32_t ranks_seen = 0;
    for (uint32_t r = 0; r < world; r++) {  /* drop the padding: rank r's shard is [rb, re) */
        uint64_t rb, re, got[4];
        CHECK_SZ(sz_shard_range(n_blocks_total, r, world, &rb, &re));
        memcpy(h_all + rb, h_pad + (size_t)r * per, (size_t)(re - rb) * 4);
        tag_of(h_all + rb, re - rb, r, got);  /* counted, not inferred: the slice that ARRIVED against the tag its rank SENT */
        ranks_seen += memcmp(got, h_tags + (size_t)r * 4, sizeof got) == 0 ? 1u : 0u;
    }
    /* layout of the whole stream; refuses the kernels' markers */
    uint64_t *offsets = (uint64_t *)malloc(((size_t)n_blocks_total + 1) * 8), bad = 0, sum = 0;
    uint64_t preamble = 0;
    for (uint64_t v = n_bytes; ; v >>= 7) {
        preamble++;
        if (v < 0x80)
            break;
    }
    const int rc = sz_stream_offsets(h_all, n_blocks_total, preamble, SZ_SLOT_STRIDE, offsets, &bad);
    for (uint64_t k = 0; k < n_blocks_total; k++)
        sum += h_all[k];
    const int slice_ok = memcmp(h_all + b, h_mine, (size_t)mine * 4) == 0;
    const int total_ok = rc == SZ_OK && offsets[n_blocks_total] == preamble + sum;

    printf("{\"rank\": %u, \"world\": %u, \"n_blocks\": %llu, \"begin\": %llu, \"end\": %llu, \"rccl_ranks_seen\": %u, "
           "\"own_slice_ok\": %s, \"layout_ok\": %s, \"stream_bytes\": %llu, \"lens\": [",
           rank, world, (unsigned long long)n_blocks_total, (unsigned long long)b, (unsigned long long)e, ranks_seen,
           slice_ok ? "true" : "false", total_ok ? "true" : "false", (unsigned long long)(rc == SZ_OK ? offsets[n_blocks_total] : 0));
    for (uint64_t k = 0; k < n_blocks_total; k++)
        printf("%s%u", k ? ", " : "", h_all[k]);
    printf("]}\n");

    g_comm_up = 0;
    ncclCommDestroy(comm);
    (void)sz_release_stream_resources(stream);
    (void)hipStreamDestroy(stream);
    return rc == SZ_OK && slice_ok && total_ok && ranks_seen == world ? 0 : 1;
}
