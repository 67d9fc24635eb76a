/* include/rocc.h -- host stand-in for the reference's tests/rocc.h: same macro names, backed by
 * the MI355X C ABI.  See rocc_host_shim.h. */
#include "rocc_host_shim.h"
