// qual_pack.h -- host helpers of the pass-bit form (qual_pack.cpp): the base-quality test of mod.rs:33 taken on the
// host, one bit per base, and the sum of the passing qualities (contig_profiler.rs:68-70).
#pragma once
#include <stdint.h>

namespace dut {

// 0 scalar, 1 SSE2, 2 AVX2: what this CPU runs (the `level` arguments below let a test pin a lower one)
int qual_pack_level();
// out[w] bit i = (q[64 w + i] >= thr), for n_words whole words (64 n_words bytes are read)
void qual_pass_words(const uint8_t *q, uint64_t n_words, uint8_t thr, uint64_t *out, int level = 2);
// the same for n <= 64 bytes: bits [n, 64) are zero
uint64_t qual_pass_partial(const uint8_t *q, uint32_t n, uint8_t thr);
// both in one pass over a read's n quality bytes: out[0, ceil(n / 64)) (zeros above bit n), returns the sum
uint64_t qual_pass_read(const uint8_t *q, uint64_t n, uint8_t thr, uint64_t *out, int level = 2);
// sum of q[i] over the i < n with q[i] >= thr
uint64_t qual_pass_sum(const uint8_t *q, uint64_t n, uint8_t thr, int level = 2);
// the reference's "is N" bits (mod.rs:100-101): bit i of out[w] = ((ref[64 w + i] | 0x20) == 'n') for the n_bases bases
// at `ref`, 1 for every position beyond them (mod.rs:79-80: no reference base reads as 'N'); n_words words are written
void ref_n_words(const uint8_t *ref, uint64_t n_bases, uint64_t n_words, uint64_t *out, int level = 2);

} // namespace dut
