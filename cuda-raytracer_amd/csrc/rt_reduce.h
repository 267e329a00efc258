/*
 * rt_reduce.h — the multi-GPU exchange step on plain lists (csrc/rt_reduce.cpp): rank i = position i in the lists.
 * rt_kernels.hip turns contexts into these lists for rt_reduce_accum / rt_reduce_scatter_rows (include/rt_api.h).
 */
#ifndef RT_REDUCE_H
#define RT_REDUCE_H
#include <hip/hip_runtime_api.h>
#include <stdint.h>

int rt_reduce_accum_lists(const int* devices, const hipStream_t* streams, float* const* accum, int n, uint64_t count, int root);
int rt_reduce_scatter_lists(const int* devices, const hipStream_t* streams, float* const* accum, int n, int w, int h, int* rows);

#endif
