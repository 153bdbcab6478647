// Shared by the Winograd kernels (conv_wino.hip: f32, conv_wino6.hip: bf16x6): the fixed accumulator registers and the
// compile-time loops.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>

namespace mn {

// ---- the 256 accumulator registers are FIXED physical AGPRs a0..a255 (position p = a[16 p : 16 p + 15]), touched only by
// inline asm.  With the MFMA builtin (or asm with "+a" operands) hipcc 7.2's allocator treats accumulators and operands as
// one either-file register class and, in this hand-ordered stream (sched_barriers between the slots), shuffles accumulator
// pieces through VGPRs and scratch every iteration (238 spills, -Rpass-analysis).  Every MFMA statement names all AGPRs as
// clobbered, so the compiler never keeps a value there; tests/test_build_resources.py holds the kernel to 0 spills (a VGPR
// spill could be parked in an AGPR between two statements) and to no compiler-generated v_accvgpr_* at all.
// The hazard recogniser does not see inside asm: the stream keeps >= 2 instructions between a VALU / LDS write of an
// operand and the MFMA that reads it (the s_waitcnt for LDS operands is still inserted by the compiler), v[15] -- read by
// the last MFMA of a step -- is rewritten one slot later, and explicit s_nops separate the last MFMA from the epilogue's
// accumulator reads and the accumulator zeroing from the next MFMA.
#define W_ACLOB \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", \
  "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", \
  "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", \
  "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", \
  "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", \
  "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", \
  "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", \
  "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", \
  "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", \
  "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", \
  "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", \
  "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", \
  "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", \
  "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", \
  "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", \
  "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", \
  "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", \
  "a253", "a254", "a255"
template <int I>
__device__ __forceinline__ float agpr_get() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(I));
  return x;
}
template <int I>
__device__ __forceinline__ void agpr_zero() {
  asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(I));
}
template <class F, int... Is>
__device__ __forceinline__ void wfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wfor(F&& f) {
  wfor_impl(f, std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ unsigned launder(unsigned x) { asm volatile("" : "+v"(x)); return x; }

}  // namespace mn
