/*
 * include/rocc_host_shim.h -- host replacement for the reference's tests/rocc.h.
 *
 * The reference drives its accelerator with RISC-V custom-3 instructions emitted through the
 * macros of tests/rocc.h:30-46 (ROCC_INSTRUCTION_S / _DSS / ...).  On an x86 host with an MI355X
 * the "instruction" is a call into the C ABI (rocc_cmd, include/snappy_hip.h).  Force-including
 * this file lets the reference's bare-metal driver build UNCHANGED:
 *
 *     gcc -std=gnu99 -include include/rocc_host_shim.h <reference>/tests/compression.c \
 *         -Lcompression-accelerator_amd/csrc -lsnappy_hip
 *
 * How it works: (1) it defines the reference header's include guard (tests/rocc.h:4-5) so the
 * RISC-V encoders are skipped; (2) it defines the same macro names with the same argument order;
 * (3) it turns `asm volatile ("fence")` (tests/compression.c:23,35) into a no-op -- hipMemcpy
 * inside the command already orders host memory against the device.
 */
#ifndef ROCC_HOST_SHIM_H
#define ROCC_HOST_SHIM_H
#define SRC_MAIN_C_ROCC_H /* include guard of the reference's tests/rocc.h */

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif
uint64_t rocc_cmd(int channel, int funct, uint64_t rs1, uint64_t rs2);
#ifdef __cplusplus
}
#endif

/* rd, rs1, rs2 are data, X is the custom-N channel, funct the 7-bit function code
 * (argument order of tests/rocc.h:30-46). */
#define ROCC_INSTRUCTION_DSS(X, rd, rs1, rs2, funct) \
    { (rd) = rocc_cmd((X), (funct), (uint64_t)(uintptr_t)(rs1), (uint64_t)(uintptr_t)(rs2)); }
#define ROCC_INSTRUCTION_DS(X, rd, rs1, funct) \
    { (rd) = rocc_cmd((X), (funct), (uint64_t)(uintptr_t)(rs1), 0); }
#define ROCC_INSTRUCTION_D(X, rd, funct) \
    { (rd) = rocc_cmd((X), (funct), 0, 0); }
#define ROCC_INSTRUCTION_SS(X, rs1, rs2, funct) \
    { (void)rocc_cmd((X), (funct), (uint64_t)(uintptr_t)(rs1), (uint64_t)(uintptr_t)(rs2)); }
#define ROCC_INSTRUCTION_S(X, rs1, funct) \
    { (void)rocc_cmd((X), (funct), (uint64_t)(uintptr_t)(rs1), 0); }
#define ROCC_INSTRUCTION(X, funct) \
    { (void)rocc_cmd((X), (funct), 0, 0); }

/* `asm volatile ("fence");` -> `(void) ("fence");`  (after the system headers above, which use
 * the real keywords) */
#define asm (void)
#define volatile

#endif /* ROCC_HOST_SHIM_H */
