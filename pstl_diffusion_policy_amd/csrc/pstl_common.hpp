// pstl_common.hpp -- shared by the HIP translation units of libpstl_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pstl_hip.h"

namespace pstl {

inline int check_cfg(const pstl_cfg* c) {
  if (!c) return PSTL_ERR_ARG;
  if (c->bs <= 0 || c->rows_per_scene <= 0 || c->K < 1 || c->steps < 2) return PSTL_ERR_ARG;  // the reference needs K >= 1 too
  return PSTL_OK;
}

// Row buffers of 40 controls / (T,4) states are read and written 16 bytes at a time: they must be 16-byte aligned (every
// torch allocation and every row-wise slice of one is).
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline long n_rows(const pstl_cfg* c) { return (long)c->bs * c->rows_per_scene; }

inline int launch_status() { return hipGetLastError() == hipSuccess ? PSTL_OK : PSTL_ERR_LAUNCH; }

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace pstl
