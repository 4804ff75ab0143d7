// Run-time binding of RCCL for the C ABI's multi-GPU entry points (see comm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <string>

namespace pylda __attribute__((visibility("hidden"))) {

int comm_unique_id(void* id_out, std::string* err);
int comm_init(void** comm, const void* id_bytes, int rank, int world, std::string* err);
void comm_destroy(void* comm);
int comm_allreduce_sum_f64(void* comm, double* device_buffer, size_t count, hipStream_t stream, std::string* err);

}  // namespace pylda
