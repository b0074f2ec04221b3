// Internal glue: the public ABI (include/tavhip.h) plus launch helpers.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tavhip.h"

#define TAV_ABI_VERSION 6

static inline int tav_last_error() { return (int)hipGetLastError(); }
static inline unsigned tav_cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }
