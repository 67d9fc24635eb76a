/*
 * rocc_compress.c -- host driver with the call shape of the reference's bare-metal program
 * (tests/compression.c): SET_LENGTH then COMPRESS through the RoCC instruction macros, then
 * SET_LENGTH + UNCOMPRESS, on an MI355X instead of a Rocket tile.
 *
 * Build:  gcc -std=gnu99 -include include/rocc_host_shim.h rocc_compress.c -lsnappy_hip
 * Usage:  rocc_compress [input-file [output-file]]     (default input: the thistle sentence)
 * Prints  `<uncompressed> <compressed> <uncompress-result>`; exit status 0 iff the round trip
 * reproduced the input.  With an output file, the compressed stream is written there.
 */
#include "rocc.h" /* replaced by include/rocc_host_shim.h on the host */

#define ACCEL_CHANNEL 3
#define FN_COMPRESS 0
#define FN_UNCOMPRESS 1
#define FN_SET_LENGTH 2

static unsigned long accel_compress(const char *src, unsigned long len, char *dst)
{
    unsigned long out;
    ROCC_INSTRUCTION_S(ACCEL_CHANNEL, len, FN_SET_LENGTH);
    ROCC_INSTRUCTION_DSS(ACCEL_CHANNEL, out, src, dst, FN_COMPRESS);
    return out;
}

static unsigned long accel_uncompress(const char *src, unsigned long len, char *dst)
{
    unsigned long failed;
    ROCC_INSTRUCTION_S(ACCEL_CHANNEL, len, FN_SET_LENGTH);
    ROCC_INSTRUCTION_DSS(ACCEL_CHANNEL, failed, src, dst, FN_UNCOMPRESS);
    return failed;
}

int main(int argc, char **argv)
{
    static const char sentence[] =
        "If Tid the thistle sifter lifts his sieve of sifted thistles, then he'll shift the thistles in "
        "his sifting sieve to live unsifted.";
    char *input;
    unsigned long n;
    if (argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        if (!f) {
            perror(argv[1]);
            return 2;
        }
        fseek(f, 0, SEEK_END);
        n = (unsigned long)ftell(f);
        fseek(f, 0, SEEK_SET);
        input = (char *)malloc(n + 1);
        if (fread(input, 1, n, f) != n) {
            fclose(f);
            return 2;
        }
        fclose(f);
    } else {
        n = sizeof(sentence) - 1;
        input = (char *)malloc(n + 1);
        memcpy(input, sentence, n);
    }
    char *packed = (char *)malloc(32 + n + n / 6);
    char *back = (char *)malloc(n + 1);
    unsigned long clen = accel_compress(input, n, packed);
    if (clen == (unsigned long)-1)
        return 3; /* no GPU / HIP failure: already reported on stderr, no CPU fallback */
    if (argc > 2) {
        FILE *o = fopen(argv[2], "wb");
        if (!o || fwrite(packed, 1, clen, o) != clen)
            return 2;
        fclose(o);
    }
    unsigned long failed = accel_uncompress(packed, clen, back);
    printf("%lu %lu %lu\n", n, clen, failed);
    return (failed == 0 && memcmp(input, back, n) == 0) ? 0 : 1;
}
