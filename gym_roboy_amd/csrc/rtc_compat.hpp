// rtc_compat.hpp - the few standard declarations the device headers use, for both compilers: hipcc (the
// library build: the real headers) and hiprtc (run-time specialisation, msj_jit.hpp: no standard library
// headers are reachable there, so the fixed-width integer names and std::is_same are declared by hand).
#pragma once
#if defined(__HIPCC_RTC__)
// the compiler's own LP64 types (what <cstdint> gives the library build: the same mangled names in both builds; and a
// hiprtc that came to pre-declare these names the standard way would meet an identical, hence legal, redeclaration)
typedef __INT8_TYPE__ int8_t;
typedef __UINT8_TYPE__ uint8_t;
typedef __INT32_TYPE__ int32_t;
typedef __UINT32_TYPE__ uint32_t;
typedef __INT64_TYPE__ int64_t;
typedef __UINT64_TYPE__ uint64_t;
typedef __UINTPTR_TYPE__ uintptr_t;
namespace std {
template <class A, class B> struct is_same { static constexpr bool value = false; };
template <class A> struct is_same<A, A> { static constexpr bool value = true; };
}  // namespace std
#else
#include <cmath>
#include <cstdint>
#include <type_traits>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#endif
