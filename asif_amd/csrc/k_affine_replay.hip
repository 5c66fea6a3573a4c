// k_affine_replay.hip -- asif_hip_affine_replay: runs a register program on the device's affine forms
// (affine_dev.hpp, the arithmetic behind the robust and realizable rows) in one lane and returns every register.
// It exists so that the parity tests can replay, on the GPU, the instruction programs whose results the
// reference's libaffa produced (tests/golden/affa_programs.json, oracle/ref_affa_shim.cpp) and compare centres,
// bounds and coefficients bit for bit -- including the operations no shipped model reaches (sin of wide or tiny
// intervals, inv of a form straddling zero).
#include "affine_dev.hpp"
#include "asif_hip.h"

namespace asif {

constexpr int kReplayMaxReg = 16;

__global__ void affine_replay_kernel(const asif_hip_affine_instr *prog, int nprog, int nreg, double *center, int32_t *n,
                                     double *lo, double *hi, uint32_t *idx, double *coef, int32_t *overflow)
{
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	Af R[kReplayMaxReg];
	AfCtx cx = {0u, false};
	for (int r = 0; r < nreg; r++) af_const(R[r], 0.0);
	for (int p = 0; p < nprog; p++) {
		const asif_hip_affine_instr I = prog[p];
		Af t; // results go through a temporary: source and destination registers may coincide
		switch (I.op) {
		case 0: af_const(t, I.imm0); break;
		case 1: af_interval(cx, t, I.imm0, I.imm1); break;
		case 2: af_add(cx, R[I.a], R[I.b], t); break;
		case 3: af_sub(cx, R[I.a], R[I.b], t); break;
		case 4: af_mul(cx, R[I.a], R[I.b], t); break;
		case 5: af_div(cx, R[I.a], R[I.b], t); break;
		case 6: af_inv(cx, R[I.a], t); break;
		case 7: af_neg(R[I.a], t); break;
		case 8: af_scale(R[I.a], I.imm0, t); break;
		case 9: af_sin(cx, R[I.a], t); break;
		default: t = R[I.a]; break; // 10: copy
		}
		R[I.dst] = t;
	}
	for (int r = 0; r < nreg; r++) {
		center[r] = R[r].c;
		n[r] = R[r].n;
		af_convert(R[r], lo[r], hi[r]);
		for (int k = 0; k < R[r].n && k < kAfCap; k++) {
			idx[r * kAfCap + k] = R[r].idx[k];
			coef[r * kAfCap + k] = R[r].v[k];
		}
	}
	*overflow = cx.overflow ? 1 : 0;
}

} // namespace asif

extern "C" int asif_hip_affine_replay(int device, const asif_hip_affine_instr *prog, int32_t nprog, int32_t nreg,
                                      double *center, int32_t *n, double *lo, double *hi, uint32_t *idx, double *coef)
{
	using namespace asif;
	if (!prog || nprog < 0 || nreg < 1 || nreg > kReplayMaxReg || !center || !n || !lo || !hi || !idx || !coef)
		return ASIF_HIP_EINVAL;
	for (int p = 0; p < nprog; p++)
		if (prog[p].op < 0 || prog[p].op > 10 || prog[p].dst < 0 || prog[p].dst >= nreg || prog[p].a < 0 ||
		    prog[p].a >= nreg || prog[p].b < 0 || prog[p].b >= nreg)
			return ASIF_HIP_EINVAL;
	hipError_t e = hipSetDevice(device);
	if (e != hipSuccess) return ASIF_HIP_ENODEVICE;
	const size_t szP = sizeof(asif_hip_affine_instr) * (size_t)(nprog > 0 ? nprog : 1), szD = sizeof(double) * nreg;
	const size_t szC = sizeof(double) * nreg * kAfCap, szI = sizeof(uint32_t) * nreg * kAfCap;
	char *buf = nullptr;
	const size_t total = szP + 3 * szD + sizeof(int32_t) * (nreg + 2) + szC + szI + 64;
	if ((e = hipMalloc((void **)&buf, total)) != hipSuccess) return (int)e;
	(void)hipMemset(buf, 0, total);
	// doubles first (alignment), then 32-bit fields, then the program
	double *dC = (double *)buf, *dLo = dC + nreg, *dHi = dLo + nreg, *dCoef = dHi + nreg;
	uint32_t *dIdx = (uint32_t *)(dCoef + (size_t)nreg * kAfCap);
	int32_t *dN = (int32_t *)(dIdx + (size_t)nreg * kAfCap), *dOv = dN + nreg;
	asif_hip_affine_instr *dP = (asif_hip_affine_instr *)(((uintptr_t)(dOv + 1) + 15) & ~(uintptr_t)15);
	if (nprog > 0 && (e = hipMemcpy(dP, prog, sizeof(asif_hip_affine_instr) * (size_t)nprog, hipMemcpyHostToDevice)) != hipSuccess) {
		(void)hipFree(buf);
		return (int)e;
	}
	hipLaunchKernelGGL(affine_replay_kernel, dim3(1), dim3(64), 0, nullptr, dP, nprog, nreg, dC, dN, dLo, dHi, dIdx, dCoef, dOv);
	e = hipGetLastError();
	int32_t ov = 0;
	if (e == hipSuccess) e = hipMemcpy(center, dC, szD, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(lo, dLo, szD, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(hi, dHi, szD, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(coef, dCoef, szC, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(idx, dIdx, szI, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(n, dN, sizeof(int32_t) * nreg, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(&ov, dOv, sizeof(ov), hipMemcpyDeviceToHost);
	(void)hipFree(buf);
	if (e != hipSuccess) return (int)e;
	return ov ? ASIF_HIP_EUNSUPPORTED : ASIF_HIP_OK; // more than 16 noise symbols in one form
}
