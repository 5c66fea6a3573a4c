// asif_utils.h -- the small dense helpers the reference publishes to its callers under this name
// (include/asif_utils.h:22-73) and its examples use in their model callbacks
// (examples/InvertedPendulum_Implicit.cpp:49,67,70): column-major `matrixMultiply`,
// `matrixVectorMultiply`, `vectorNorm`, templated on the scalar so that affine forms (AAF) work too, plus the
// standard headers that file pulls in for its includers (<cstring>, <vector>, <functional>, <cmath>, <algorithm>).
// Accumulation order is the reference's: element (i, j) starts at T(0.0) and adds k = 0, 1, ... in turn.
#pragma once
#include <assert.h>
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>
#include "qpwrappers.h"

namespace ASIF {

// AB (nlA x ncB) = A (nlA x ncA) * B (nlB x ncB), all column-major; ncA must equal nlB
template <class T>
inline void matrixMultiply(const T A[], const uint32_t nlA, const uint32_t ncA, const T B[], const uint32_t nlB,
                           const uint32_t ncB, T AB[])
{
	assert(ncA == nlB);
	(void)nlB;
	for (uint32_t j = 0; j < ncB; j++)
		for (uint32_t i = 0; i < nlA; i++) {
			T acc = 0.0;
			for (uint32_t k = 0; k < ncA; k++) acc = acc + A[i + k * nlA] * B[k + j * ncA];
			AB[i + j * nlA] = acc;
		}
}

// Ab (nlA) = A (nlA x ncA, column-major) * b (nlb); ncA must equal nlb
template <class T>
inline void matrixVectorMultiply(const T A[], const uint32_t nlA, const uint32_t ncA, const T b[], const uint32_t nlb,
                                 T Ab[])
{
	assert(ncA == nlb);
	(void)nlb;
	for (uint32_t i = 0; i < nlA; i++) {
		T acc = 0.0;
		for (uint32_t k = 0; k < ncA; k++) acc = acc + A[i + k * nlA] * b[k];
		Ab[i] = acc;
	}
}

// Euclidean norm; the sum of squares is accumulated in double whatever T is, as in the reference
template <class T>
inline T vectorNorm(const T v[], const uint32_t len)
{
	double sumsq = 0;
	for (uint32_t i = 0; i < len; i++) sumsq += v[i] * v[i];
	return sqrt(sumsq);
}

} // namespace ASIF
