/* csrc/nbody_error.c -- error convention of the C ABI (include/nbody.h): no exceptions, no exit().
 * Replaces the reference's CUDA_SYNC_CHECK throw (src/nbody.cu:20-33) and its exit(0/1) paths
 * (src/nbody.cu:68-72, include/nbodyConfig.h:25-28). */
#include "nbody.h"
#include "nbody_error.h"
#include <stdarg.h>
#include <stdio.h>

static __thread char t_last_error[512] = "no error";

int nbody_fail(int status, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_last_error, sizeof(t_last_error), fmt, ap);
    va_end(ap);
    return status;
}

const char* nbody_last_error_string(void) { return t_last_error; }

const char* nbody_status_string(int status) {
    switch (status) {
        case NBODY_OK: return "NBODY_OK";
        case NBODY_ERR_INVALID: return "NBODY_ERR_INVALID";
        case NBODY_ERR_IO: return "NBODY_ERR_IO";
        case NBODY_ERR_PARSE: return "NBODY_ERR_PARSE";
        case NBODY_ERR_NOMEM: return "NBODY_ERR_NOMEM";
        case NBODY_ERR_NO_DEVICE: return "NBODY_ERR_NO_DEVICE";
        case NBODY_ERR_HIP: return "NBODY_ERR_HIP";
        case NBODY_ERR_CAPACITY: return "NBODY_ERR_CAPACITY";
        case NBODY_ERR_COMM: return "NBODY_ERR_COMM";
        case NBODY_ERR_STATE: return "NBODY_ERR_STATE";
        default: return "NBODY_ERR_UNKNOWN";
    }
}

int nbody_abi_version(void) { return NBODY_ABI_VERSION; }
