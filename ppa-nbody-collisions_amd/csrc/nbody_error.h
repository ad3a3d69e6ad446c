/* csrc/nbody_error.h -- thread-local last-error plumbing shared by the C host code and the HIP side. */
#ifndef NBODY_ERROR_H
#define NBODY_ERROR_H
#ifdef __cplusplus
extern "C" {
#endif
/* Records a printf-formatted message for nbody_last_error_string() and returns `status`. */
int nbody_fail(int status, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#ifdef __cplusplus
}
#endif
#endif
