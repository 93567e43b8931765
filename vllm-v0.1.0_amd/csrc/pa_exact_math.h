// Arithmetic that rounds like the reference's element types: c10::Half and
// c10::BFloat16 round the result of every operator to the storage type.  Files
// that include this are built without fast-math and with floating-point
// contraction off (see the Makefile), so that a * b - c * d stays two products
// and a difference for float as well.
#pragma once

#include "pa_common.h"

namespace pa {

template <typename T> struct Num;
template <> struct Num<float> {
  static __device__ __forceinline__ float to(float x) { return x; }
  static __device__ __forceinline__ float from(float x) { return x; }
};
template <> struct Num<_Float16> {
  static __device__ __forceinline__ float to(_Float16 x) { return (float)x; }
  static __device__ __forceinline__ _Float16 from(float x) { return (_Float16)x; }
};
template <> struct Num<__bf16> {
  static __device__ __forceinline__ float to(__bf16 x) { return (float)x; }
  static __device__ __forceinline__ __bf16 from(float x) { return (__bf16)x; }
};

// a op b with the result rounded to T, as c10::Half / c10::BFloat16 operators do
template <typename T> __device__ __forceinline__ T mul_t(T a, T b) { return Num<T>::from(Num<T>::to(a) * Num<T>::to(b)); }
template <typename T> __device__ __forceinline__ T add_t(T a, T b) { return Num<T>::from(Num<T>::to(a) + Num<T>::to(b)); }
template <typename T> __device__ __forceinline__ T sub_t(T a, T b) { return Num<T>::from(Num<T>::to(a) - Num<T>::to(b)); }

template <typename T, int N>
struct alignas(16) Pack {
  T v[N];
};

}  // namespace pa
