// Host-side context shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <string>

#include "../../include/ammsb.h"

struct ammsb_ctx {
  int device;
  ammsb_params params;
  int num_cus;
  // reduction workspace (allocated once in ammsb_ctx_create; never reallocated afterwards)
  float* grad_partials;      // [max_partials, 2K]
  uint32_t max_partials;
  float* theta_sum;          // [K]   (BetaUpdater::GetThetaSum())
  float4* theta_coef;        // [K]   per-column constants of the gradient kernels (theta_coef, ammsb_beta.hip)
  double* ppx_partials;      // [max_ppx_blocks, 2]
  unsigned long long* ppx_cnt_partials;  // [max_ppx_blocks, 2]
  uint32_t max_ppx_blocks;
  uint32_t* ppx_ticket;      // [1]   blocks of a self-reducing perplexity launch that have written their partials (0 between launches)
  // name of the kernel the last update_phi / update_pi / beta gradient / perplexity call dispatched to, as the
  // rocprofv3 kernel trace spells it (ammsb_last_kernel_name)
  const char* kernel_name[5];
  char err[256];
};

enum { AMMSB_KN_PHI = 0, AMMSB_KN_PI = 1, AMMSB_KN_GRADS = 2, AMMSB_KN_PPX = 3, AMMSB_KN_PHI_SMALL = 4 };

// "kernel<template arguments>" as the demangler prints it; one static string per launcher instantiation
static inline std::string ammsb_kname(const char* fmt, ...) {
  char buf[128];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return std::string(buf);
}

#define AMMSB_CHECK_ARG(ctx, cond, msg)                                   \
  do {                                                                    \
    if (!(cond)) {                                                        \
      if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), "%s: %s", __func__, msg); \
      return AMMSB_EINVAL;                                                \
    }                                                                     \
  } while (0)

#define AMMSB_HIP(ctx, call)                                                                   \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), "%s: %s -> %s", __func__, #call,       \
                        hipGetErrorString(e_));                                                \
      return AMMSB_EHIP;                                                                       \
    }                                                                                          \
  } while (0)

#define AMMSB_LAUNCH_CHECK(ctx) AMMSB_HIP(ctx, hipGetLastError())

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// dispatch a power-of-two work-group size to a template instantiation
#define AMMSB_DISPATCH_L(wg, ...)                      \
  switch (wg) {                                        \
    case 1: { constexpr int L_ = 1; __VA_ARGS__; } break;     \
    case 2: { constexpr int L_ = 2; __VA_ARGS__; } break;     \
    case 4: { constexpr int L_ = 4; __VA_ARGS__; } break;     \
    case 8: { constexpr int L_ = 8; __VA_ARGS__; } break;     \
    case 16: { constexpr int L_ = 16; __VA_ARGS__; } break;   \
    case 32: { constexpr int L_ = 32; __VA_ARGS__; } break;   \
    case 64: { constexpr int L_ = 64; __VA_ARGS__; } break;   \
    case 128: { constexpr int L_ = 128; __VA_ARGS__; } break; \
    case 256: { constexpr int L_ = 256; __VA_ARGS__; } break; \
    case 512: { constexpr int L_ = 512; __VA_ARGS__; } break; \
    case 1024: { constexpr int L_ = 1024; __VA_ARGS__; } break; \
    default: return AMMSB_EINVAL;                      \
  }
