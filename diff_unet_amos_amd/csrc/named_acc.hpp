// MFMA accumulators held in the upper half of the vector registers, v[128:255], BY NAME.
//
// Why: a kernel whose eight 32x32 accumulator tuples live across several loops (half chunks x phases, then a second K section,
// or a tile loop) hands hipcc 128 registers of loop-carried tuples to move around -- it copies them at loop exits and spills
// 100-230 registers (profiles/r4_conv_wide_persistent_tiles_ab.txt; the first form of upconv.hip: 136 in straight-line code).
// Written literally, the accumulators are not values of the program at all: an MFMA statement has two VGPR inputs and no
// output, so the register allocator sees a kernel of ~120 live registers and the accumulators stay where they are.
//
// How the registers are kept ours (cdna_hip_programming.md 5.7 item 4):
//   * the kernel carries DUA_NAMED_ACC_KERNEL (amdgpu_num_vgpr): the allocator may use v0..v127 only, v128..v255 are reserved
//     registers to it; every statement below lists the registers it touches as clobbers, which is what makes the kernel descriptor
//     allocate all 256 (two waves per SIMD: 2 x 256 of the 512-entry file);
//   * no statement of the kernel has an "a"-class operand, so the compiler does not plan with accumulator registers at all
//     (the first attempt kept the tuples in a[0:127]: hipcc allocates vector values to free AGPRs on gfx950 -- copies through
//     a0..a2 appeared between two MFMA statements; there is no attribute that takes the AGPRs away from it);
//   * tools/audit_named_acc.sh (run by the Makefile) checks the ISA of every kernel that includes this header: no access to
//     v128..v255 outside ASMSTART / ASMEND, no AGPRs, no scratch.
// Wait states are ours (nothing inside an asm statement is padded): a v_mov into an accumulator -> the MFMA reading it as C
// (named_acc_fence_init), the last MFMA -> a v_mov out of its destination (named_acc_fence_read: 32 states; a 32x32x16 MFMA
// is 8 passes).  An MFMA that takes the previous MFMA's D whole as its C needs none.
#pragma once
#include "common.hpp"

// gfx950 doubles the attribute's number (it counts the unified VGPR + AGPR file): 64 leaves the allocator v0..v127
#define DUA_NAMED_ACC_KERNEL __attribute__((amdgpu_num_vgpr(64)))
// "inline asm clobber list contains reserved registers: v128 ...": that they are reserved is the point
#pragma clang diagnostic ignored "-Winline-asm"

namespace dua {

// accumulator tuple I (0..7) = v[128 + 16 I : 128 + 16 I + 15]
template <int I> __device__ __forceinline__ void named_mfma(const f16x8& a, const f16x8& b);
template <int I> __device__ __forceinline__ void named_write16(const float (&v)[16]);
template <int I> __device__ __forceinline__ void named_read16(float (&v)[16]);
template <> __device__ __forceinline__ void named_mfma<0>(const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 v[128:143], %0, %1, v[128:143]" ::"v"(a), "v"(b) : "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
}
template <> __device__ __forceinline__ void named_write16<0>(const float (&v)[16]) {
  asm volatile("v_mov_b32 v128, %0\n\tv_mov_b32 v129, %1\n\tv_mov_b32 v130, %2\n\tv_mov_b32 v131, %3\n\tv_mov_b32 v132, %4\n\tv_mov_b32 v133, %5\n\tv_mov_b32 v134, %6\n\tv_mov_b32 v135, %7\n\tv_mov_b32 v136, %8\n\tv_mov_b32 v137, %9\n\tv_mov_b32 v138, %10\n\tv_mov_b32 v139, %11\n\tv_mov_b32 v140, %12\n\tv_mov_b32 v141, %13\n\tv_mov_b32 v142, %14\n\tv_mov_b32 v143, %15" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
}
template <> __device__ __forceinline__ void named_read16<0>(float (&v)[16]) {
  asm volatile("v_mov_b32 %0, v128\n\tv_mov_b32 %1, v129\n\tv_mov_b32 %2, v130\n\tv_mov_b32 %3, v131\n\tv_mov_b32 %4, v132\n\tv_mov_b32 %5, v133\n\tv_mov_b32 %6, v134\n\tv_mov_b32 %7, v135\n\tv_mov_b32 %8, v136\n\tv_mov_b32 %9, v137\n\tv_mov_b32 %10, v138\n\tv_mov_b32 %11, v139\n\tv_mov_b32 %12, v140\n\tv_mov_b32 %13, v141\n\tv_mov_b32 %14, v142\n\tv_mov_b32 %15, v143" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
}
template <> __device__ __forceinline__ void named_mfma<1>(const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 v[144:159], %0, %1, v[144:159]" ::"v"(a), "v"(b) : "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159");
}
template <> __device__ __forceinline__ void named_write16<1>(const float (&v)[16]) {
  asm volatile("v_mov_b32 v144, %0\n\tv_mov_b32 v145, %1\n\tv_mov_b32 v146, %2\n\tv_mov_b32 v147, %3\n\tv_mov_b32 v148, %4\n\tv_mov_b32 v149, %5\n\tv_mov_b32 v150, %6\n\tv_mov_b32 v151, %7\n\tv_mov_b32 v152, %8\n\tv_mov_b32 v153, %9\n\tv_mov_b32 v154, %10\n\tv_mov_b32 v155, %11\n\tv_mov_b32 v156, %12\n\tv_mov_b32 v157, %13\n\tv_mov_b32 v158, %14\n\tv_mov_b32 v159, %15" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159");
}
template <> __device__ __forceinline__ void named_read16<1>(float (&v)[16]) {
  asm volatile("v_mov_b32 %0, v144\n\tv_mov_b32 %1, v145\n\tv_mov_b32 %2, v146\n\tv_mov_b32 %3, v147\n\tv_mov_b32 %4, v148\n\tv_mov_b32 %5, v149\n\tv_mov_b32 %6, v150\n\tv_mov_b32 %7, v151\n\tv_mov_b32 %8, v152\n\tv_mov_b32 %9, v153\n\tv_mov_b32 %10, v154\n\tv_mov_b32 %11, v155\n\tv_mov_b32 %12, v156\n\tv_mov_b32 %13, v157\n\tv_mov_b32 %14, v158\n\tv_mov_b32 %15, v159" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
}
template <> __device__ __forceinline__ void named_mfma<2>(const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 v[160:175], %0, %1, v[160:175]" ::"v"(a), "v"(b) : "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175");
}
template <> __device__ __forceinline__ void named_write16<2>(const float (&v)[16]) {
  asm volatile("v_mov_b32 v160, %0\n\tv_mov_b32 v161, %1\n\tv_mov_b32 v162, %2\n\tv_mov_b32 v163, %3\n\tv_mov_b32 v164, %4\n\tv_mov_b32 v165, %5\n\tv_mov_b32 v166, %6\n\tv_mov_b32 v167, %7\n\tv_mov_b32 v168, %8\n\tv_mov_b32 v169, %9\n\tv_mov_b32 v170, %10\n\tv_mov_b32 v171, %11\n\tv_mov_b32 v172, %12\n\tv_mov_b32 v173, %13\n\tv_mov_b32 v174, %14\n\tv_mov_b32 v175, %15" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175");
}
template <> __device__ __forceinline__ void named_read16<2>(float (&v)[16]) {
  asm volatile("v_mov_b32 %0, v160\n\tv_mov_b32 %1, v161\n\tv_mov_b32 %2, v162\n\tv_mov_b32 %3, v163\n\tv_mov_b32 %4, v164\n\tv_mov_b32 %5, v165\n\tv_mov_b32 %6, v166\n\tv_mov_b32 %7, v167\n\tv_mov_b32 %8, v168\n\tv_mov_b32 %9, v169\n\tv_mov_b32 %10, v170\n\tv_mov_b32 %11, v171\n\tv_mov_b32 %12, v172\n\tv_mov_b32 %13, v173\n\tv_mov_b32 %14, v174\n\tv_mov_b32 %15, v175" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
}
template <> __device__ __forceinline__ void named_mfma<3>(const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 v[176:191], %0, %1, v[176:191]" ::"v"(a), "v"(b) : "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191");
}
template <> __device__ __forceinline__ void named_write16<3>(const float (&v)[16]) {
  asm volatile("v_mov_b32 v176, %0\n\tv_mov_b32 v177, %1\n\tv_mov_b32 v178, %2\n\tv_mov_b32 v179, %3\n\tv_mov_b32 v180, %4\n\tv_mov_b32 v181, %5\n\tv_mov_b32 v182, %6\n\tv_mov_b32 v183, %7\n\tv_mov_b32 v184, %8\n\tv_mov_b32 v185, %9\n\tv_mov_b32 v186, %10\n\tv_mov_b32 v187, %11\n\tv_mov_b32 v188, %12\n\tv_mov_b32 v189, %13\n\tv_mov_b32 v190, %14\n\tv_mov_b32 v191, %15" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191");
}
template <> __device__ __forceinline__ void named_read16<3>(float (&v)[16]) {
  asm volatile("v_mov_b32 %0, v176\n\tv_mov_b32 %1, v177\n\tv_mov_b32 %2, v178\n\tv_mov_b32 %3, v179\n\tv_mov_b32 %4, v180\n\tv_mov_b32 %5, v181\n\tv_mov_b32 %6, v182\n\tv_mov_b32 %7, v183\n\tv_mov_b32 %8, v184\n\tv_mov_b32 %9, v185\n\tv_mov_b32 %10, v186\n\tv_mov_b32 %11, v187\n\tv_mov_b32 %12, v188\n\tv_mov_b32 %13, v189\n\tv_mov_b32 %14, v190\n\tv_mov_b32 %15, v191" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
}
template <> __device__ __forceinline__ void named_mfma<4>(const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 v[192:207], %0, %1, v[192:207]" ::"v"(a), "v"(b) : "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207");
}
template <> __device__ __forceinline__ void named_write16<4>(const float (&v)[16]) {
  asm volatile("v_mov_b32 v192, %0\n\tv_mov_b32 v193, %1\n\tv_mov_b32 v194, %2\n\tv_mov_b32 v195, %3\n\tv_mov_b32 v196, %4\n\tv_mov_b32 v197, %5\n\tv_mov_b32 v198, %6\n\tv_mov_b32 v199, %7\n\tv_mov_b32 v200, %8\n\tv_mov_b32 v201, %9\n\tv_mov_b32 v202, %10\n\tv_mov_b32 v203, %11\n\tv_mov_b32 v204, %12\n\tv_mov_b32 v205, %13\n\tv_mov_b32 v206, %14\n\tv_mov_b32 v207, %15" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207");
}
template <> __device__ __forceinline__ void named_read16<4>(float (&v)[16]) {
  asm volatile("v_mov_b32 %0, v192\n\tv_mov_b32 %1, v193\n\tv_mov_b32 %2, v194\n\tv_mov_b32 %3, v195\n\tv_mov_b32 %4, v196\n\tv_mov_b32 %5, v197\n\tv_mov_b32 %6, v198\n\tv_mov_b32 %7, v199\n\tv_mov_b32 %8, v200\n\tv_mov_b32 %9, v201\n\tv_mov_b32 %10, v202\n\tv_mov_b32 %11, v203\n\tv_mov_b32 %12, v204\n\tv_mov_b32 %13, v205\n\tv_mov_b32 %14, v206\n\tv_mov_b32 %15, v207" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
}
template <> __device__ __forceinline__ void named_mfma<5>(const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 v[208:223], %0, %1, v[208:223]" ::"v"(a), "v"(b) : "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223");
}
template <> __device__ __forceinline__ void named_write16<5>(const float (&v)[16]) {
  asm volatile("v_mov_b32 v208, %0\n\tv_mov_b32 v209, %1\n\tv_mov_b32 v210, %2\n\tv_mov_b32 v211, %3\n\tv_mov_b32 v212, %4\n\tv_mov_b32 v213, %5\n\tv_mov_b32 v214, %6\n\tv_mov_b32 v215, %7\n\tv_mov_b32 v216, %8\n\tv_mov_b32 v217, %9\n\tv_mov_b32 v218, %10\n\tv_mov_b32 v219, %11\n\tv_mov_b32 v220, %12\n\tv_mov_b32 v221, %13\n\tv_mov_b32 v222, %14\n\tv_mov_b32 v223, %15" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223");
}
template <> __device__ __forceinline__ void named_read16<5>(float (&v)[16]) {
  asm volatile("v_mov_b32 %0, v208\n\tv_mov_b32 %1, v209\n\tv_mov_b32 %2, v210\n\tv_mov_b32 %3, v211\n\tv_mov_b32 %4, v212\n\tv_mov_b32 %5, v213\n\tv_mov_b32 %6, v214\n\tv_mov_b32 %7, v215\n\tv_mov_b32 %8, v216\n\tv_mov_b32 %9, v217\n\tv_mov_b32 %10, v218\n\tv_mov_b32 %11, v219\n\tv_mov_b32 %12, v220\n\tv_mov_b32 %13, v221\n\tv_mov_b32 %14, v222\n\tv_mov_b32 %15, v223" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
}
template <> __device__ __forceinline__ void named_mfma<6>(const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 v[224:239], %0, %1, v[224:239]" ::"v"(a), "v"(b) : "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239");
}
template <> __device__ __forceinline__ void named_write16<6>(const float (&v)[16]) {
  asm volatile("v_mov_b32 v224, %0\n\tv_mov_b32 v225, %1\n\tv_mov_b32 v226, %2\n\tv_mov_b32 v227, %3\n\tv_mov_b32 v228, %4\n\tv_mov_b32 v229, %5\n\tv_mov_b32 v230, %6\n\tv_mov_b32 v231, %7\n\tv_mov_b32 v232, %8\n\tv_mov_b32 v233, %9\n\tv_mov_b32 v234, %10\n\tv_mov_b32 v235, %11\n\tv_mov_b32 v236, %12\n\tv_mov_b32 v237, %13\n\tv_mov_b32 v238, %14\n\tv_mov_b32 v239, %15" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239");
}
template <> __device__ __forceinline__ void named_read16<6>(float (&v)[16]) {
  asm volatile("v_mov_b32 %0, v224\n\tv_mov_b32 %1, v225\n\tv_mov_b32 %2, v226\n\tv_mov_b32 %3, v227\n\tv_mov_b32 %4, v228\n\tv_mov_b32 %5, v229\n\tv_mov_b32 %6, v230\n\tv_mov_b32 %7, v231\n\tv_mov_b32 %8, v232\n\tv_mov_b32 %9, v233\n\tv_mov_b32 %10, v234\n\tv_mov_b32 %11, v235\n\tv_mov_b32 %12, v236\n\tv_mov_b32 %13, v237\n\tv_mov_b32 %14, v238\n\tv_mov_b32 %15, v239" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
}
template <> __device__ __forceinline__ void named_mfma<7>(const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 v[240:255], %0, %1, v[240:255]" ::"v"(a), "v"(b) : "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
}
template <> __device__ __forceinline__ void named_write16<7>(const float (&v)[16]) {
  asm volatile("v_mov_b32 v240, %0\n\tv_mov_b32 v241, %1\n\tv_mov_b32 v242, %2\n\tv_mov_b32 v243, %3\n\tv_mov_b32 v244, %4\n\tv_mov_b32 v245, %5\n\tv_mov_b32 v246, %6\n\tv_mov_b32 v247, %7\n\tv_mov_b32 v248, %8\n\tv_mov_b32 v249, %9\n\tv_mov_b32 v250, %10\n\tv_mov_b32 v251, %11\n\tv_mov_b32 v252, %12\n\tv_mov_b32 v253, %13\n\tv_mov_b32 v254, %14\n\tv_mov_b32 v255, %15" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
}
template <> __device__ __forceinline__ void named_read16<7>(float (&v)[16]) {
  asm volatile("v_mov_b32 %0, v240\n\tv_mov_b32 %1, v241\n\tv_mov_b32 %2, v242\n\tv_mov_b32 %3, v243\n\tv_mov_b32 %4, v244\n\tv_mov_b32 %5, v245\n\tv_mov_b32 %6, v246\n\tv_mov_b32 %7, v247\n\tv_mov_b32 %8, v248\n\tv_mov_b32 %9, v249\n\tv_mov_b32 %10, v250\n\tv_mov_b32 %11, v251\n\tv_mov_b32 %12, v252\n\tv_mov_b32 %13, v253\n\tv_mov_b32 %14, v254\n\tv_mov_b32 %15, v255" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
}

// `idx` must fold to a constant (an unrolled loop index): the switch disappears
#define DUA_SEL8(fn, ...)            \
  switch (idx) {                     \
    case 0: fn<0>(__VA_ARGS__); break; \
    case 1: fn<1>(__VA_ARGS__); break; \
    case 2: fn<2>(__VA_ARGS__); break; \
    case 3: fn<3>(__VA_ARGS__); break; \
    case 4: fn<4>(__VA_ARGS__); break; \
    case 5: fn<5>(__VA_ARGS__); break; \
    case 6: fn<6>(__VA_ARGS__); break; \
    default: fn<7>(__VA_ARGS__); break; \
  }
__device__ __forceinline__ void named_mfma_sel(int idx, const f16x8& a, const f16x8& b) { DUA_SEL8(named_mfma, a, b) }
__device__ __forceinline__ void named_write16_sel(int idx, const float (&v)[16]) { DUA_SEL8(named_write16, v) }
__device__ __forceinline__ void named_read16_sel(int idx, float (&v)[16]) { DUA_SEL8(named_read16, v) }
#undef DUA_SEL8

// between the accumulator writes and the first MFMA that reads them as C
__device__ __forceinline__ void named_acc_fence_init() { asm volatile("s_nop 7"); }
// between the last MFMA and the first read of its destination
__device__ __forceinline__ void named_acc_fence_read() { asm volatile("s_nop 15\n\ts_nop 15"); }

}  // namespace dua
