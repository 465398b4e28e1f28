// step_amd/csrc/pool_vec.h -- 16-byte channel vectors and their element-wise max in the storage type (the pool kernels of pool.hip).
#pragma once
#include "common.h"

namespace step {

template <typename T, int V> struct Vec16;
template <> struct Vec16<float, 4> {
    typedef f32x4 raw;
    __device__ static __forceinline__ void unpack(const raw& r, float (&f)[4]) { for (int i = 0; i < 4; ++i) f[i] = r[i]; }
    __device__ static __forceinline__ raw pack(const float (&f)[4]) { raw r = {f[0], f[1], f[2], f[3]}; return r; }
};
template <typename T> struct Vec16h {
    typedef u16x8 raw;
    __device__ static __forceinline__ void unpack(const raw& r, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { T e; e.v = r[i]; f[i] = elem<T>::to_f32(e); }
    }
    __device__ static __forceinline__ raw pack(const float (&f)[8]) {
        raw r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = elem<T>::from_f32(f[i]).v;
        return r;
    }
};
template <> struct Vec16<bf16_t, 8> : Vec16h<bf16_t> {};
template <> struct Vec16<f16_t, 8> : Vec16h<f16_t> {};


// element-wise max of two 16-byte channel vectors in the storage type (the max of two representable
// values is representable: no rounding anywhere in a max pool).  enc / dec map a vector to and from the
// form the max is taken in.
template <typename T> struct VecMax;
template <> struct VecMax<float> {
    __device__ static __forceinline__ f32x4 enc(const f32x4& a) { return a; }
    __device__ static __forceinline__ f32x4 dec(const f32x4& a) { return a; }
    __device__ static __forceinline__ f32x4 lowest() { const float m = -__builtin_inff(); f32x4 r = {m, m, m, m}; return r; }
    __device__ static __forceinline__ f32x4 max(const f32x4& a, const f32x4& b) {
        f32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = fmaxf(a[i], b[i]);
        return r;
    }
};
#ifndef STEP_EMUL
template <> struct VecMax<f16_t> {
    __device__ static __forceinline__ u16x8 enc(const u16x8& a) { return a; }
    __device__ static __forceinline__ u16x8 dec(const u16x8& a) { return a; }
    __device__ static __forceinline__ u16x8 lowest() { u16x8 r; for (int i = 0; i < 8; ++i) r[i] = 0xfc00; return r; }   // -inf
    __device__ static __forceinline__ u16x8 max(const u16x8& a, const u16x8& b) {          // v_pk_max_f16
        return __builtin_bit_cast(u16x8, __builtin_elementwise_max(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b)));
    }
};
template <> struct VecMax<bf16_t> {
    // gfx950 has no packed bf16 max, and widening to fp32 costs ~9 VALU ops per pair (measured: the pool was
    // VALU-bound).  A sign-magnitude float orders like the two's-complement integer  x ^ ((x >> 15) & 0x7fff)
    // (negative values get their magnitude bits flipped), so planes are re-keyed once when they enter LDS (3 ops
    // per pair), every max is one v_pk_max_i16 per pair, and the same map brings the result back before the
    // store.  0 keeps the key 0; -0 < +0; NaN orders beyond +-inf, so a positive NaN propagates.
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    __device__ static __forceinline__ u16x8 enc(const u16x8& a) {
        const s16x8 sa = __builtin_bit_cast(s16x8, a);
        return __builtin_bit_cast(u16x8, sa ^ ((sa >> 15) & (short)0x7fff));
    }
    __device__ static __forceinline__ u16x8 dec(const u16x8& k) { return enc(k); }
    __device__ static __forceinline__ u16x8 lowest() { u16x8 r; for (int i = 0; i < 8; ++i) r[i] = 0x8000; return r; }   // below every key
    __device__ static __forceinline__ u16x8 max(const u16x8& a, const u16x8& b) {
        return __builtin_bit_cast(u16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b)));
    }
};
#else
template <typename T> struct VecMax {
    __device__ static inline u16x8 enc(const u16x8& a) { return a; }
    __device__ static inline u16x8 dec(const u16x8& a) { return a; }
    __device__ static inline u16x8 lowest() {                                  // -inf in the storage type
        float f[8];
        for (int i = 0; i < 8; ++i) f[i] = -__builtin_inff();
        return Vec16<T, 8>::pack(f);
    }
    __device__ static inline u16x8 max(const u16x8& a, const u16x8& b) {
        float fa[8], fb[8];
        Vec16<T, 8>::unpack(a, fa); Vec16<T, 8>::unpack(b, fb);
        for (int i = 0; i < 8; ++i) fa[i] = fmaxf(fa[i], fb[i]);
        return Vec16<T, 8>::pack(fa);
    }
};
#endif

}  // namespace step
