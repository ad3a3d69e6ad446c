/* csrc/nbody_partition.h -- the partition rule, shared by the C host code (nbody_partition, csrc/nbody_bodies.c) and
 * the device code (unpack_slots re-draws it every step, csrc/nbody_kernels.hpp).
 *
 * N bodies over `world` ranks: whole reference blocks (128 bodies = THREADS_PER_BLOCK, src/nbody.cu:36), as evenly as the
 * block count allows, in rank order.  Block-aligned own ranges keep every ring / lane group of the force kernels on
 * bodies of one rank, and re-drawing them from the survivor count after every step keeps the ranks level when bodies are
 * deleted (the reference compacts globally, src/nbody.cu:488-510). */
#ifndef NBODY_PARTITION_H
#define NBODY_PARTITION_H

#ifdef __HIP__
#define NBODY_HOST_DEVICE __host__ __device__
#else
#define NBODY_HOST_DEVICE
#endif

enum { NBODY_BLOCK = 128 };

NBODY_HOST_DEVICE static inline void nbody_own_range_of(int n, int rank, int world, int* lo, int* cnt) {
    const long long blocks = ((long long)n + NBODY_BLOCK - 1) / NBODY_BLOCK;
    long long first = blocks * rank / world * NBODY_BLOCK, last = blocks * (rank + 1) / world * NBODY_BLOCK;
    if (first > n) first = n;
    if (last > n) last = n;
    *lo = (int)first;
    *cnt = (int)(last - first);
}

/* Upper bound of any rank's own count when the body count is at most n: a rank's range can grow by a block while the
 * total shrinks, but never beyond ceil(blocks / world) blocks. */
NBODY_HOST_DEVICE static inline int nbody_own_upper_of(int n, int world) {
    const long long blocks = ((long long)n + NBODY_BLOCK - 1) / NBODY_BLOCK;
    return (int)((blocks + world - 1) / world) * NBODY_BLOCK;
}

#endif /* NBODY_PARTITION_H */
