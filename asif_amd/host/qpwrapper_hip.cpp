// qpwrapper_hip.cpp -- see include/qpwrapper_hip.h.  Host glue only: one pinned staging block, one upload, one
// C-ABI call, one download per solve().
#include "qpwrapper_hip.h"
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include <cstring>

namespace ASIF {

QPWrapperHip::QPWrapperHip(const uint32_t nv, const uint32_t nc, const bool diagonalCost, int device)
    : QPWrapperAbstract(nv, nc, diagonalCost), device_(device), host_(nullptr), dev_(nullptr), stream_(nullptr),
      status_(0), iters_(0), error_(0)
{
	// off unless asked for (include/qpwrapper_hip.h: why); ASIF_HIP_QP_WARM=1 switches it on without a rebuild
	const char *w = std::getenv("ASIF_HIP_QP_WARM");
	warmStart = w && w[0] == '1';
	haveWarm_ = false;
	asif_hip_default_solver(&settings);
	be8_.assign(nc_ > 0 ? nc_ : 1, 0);
}

QPWrapperHip::~QPWrapperHip(void)
{
	if (host_) (void)hipHostFree(host_);
	if (stream_) (void)hipStreamDestroy((hipStream_t)stream_);
}

int QPWrapperHip::setup(void)
{
	if (host_) return 0;
	hipError_t e = hipSetDevice(device_);
	if (e != hipSuccess) return ASIF_HIP_ENODEVICE;
	if ((e = hipHostMalloc((void **)&host_, sizeof(double) * total(), hipHostMallocDefault)) != hipSuccess) {
		host_ = nullptr;
		return ASIF_HIP_ENODEVICE;
	}
	std::memset(host_, 0, sizeof(double) * total());
	// the problem is a few hundred bytes: the kernel reads it from, and writes the solution to, the page-locked block
	// itself (its device-side address) -- a solve is one launch and one synchronisation, no copies
	if ((e = hipHostGetDevicePointer((void **)&dev_, host_, 0)) != hipSuccess) return (int)e;
	hipStream_t s;
	if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) return (int)e;
	stream_ = (void *)s;
	return 0;
}

int32_t QPWrapperHip::initialize(const double H[], const double c[], const double A[], const double b[],
                                 const double lb[], const double ub[], const bool be[])
{
	if ((error_ = setup()) != 0) return error_;
	haveWarm_ = false; // osqp_setup: a fresh workspace, its first solve is cold
	if (be != nullptr)
		for (uint32_t i = 0; i < nc_; i++) be_[i] = be[i];
	for (uint32_t i = 0; i < nc_; i++) be8_[i] = be_[i] ? 1 : 0;
	updateCost(H, c);
	updateA(A);
	updateb(b);
	updateBounds(lb, ub);
	// the reference solves once at the end of every initialize() and ignores the outcome (src/asif.cpp:101-105);
	// here that first solve also validates the shape against the kernels
	(void)solve();
	return error_;
}

int32_t QPWrapperHip::updateCost(const double H[], const double c[])
{
	if (!host_ && setup() != 0) return 1;
	if (H != nullptr) {
		if (diagonalCost_) // only the diagonal is read, src/qpwrapper_osqp.cpp:267-272
			for (uint32_t i = 0; i < nv_; i++) host_[offH() + i] = H[i + i * nv_];
		else // the whole matrix, of which the kernel reads the upper triangle (:136-153,276-309)
			std::memcpy(&host_[offH()], H, sizeof(double) * nv_ * nv_);
	}
	if (c != nullptr) std::memcpy(&host_[offC()], c, sizeof(double) * nv_);
	return 1;
}

int32_t QPWrapperHip::updateA(const double A[])
{
	if (!host_ && setup() != 0) return 1;
	std::memcpy(&host_[offA()], A, sizeof(double) * nc_ * nv_);
	return 1;
}

int32_t QPWrapperHip::updateb(const double b[])
{
	if (!host_ && setup() != 0) return 1;
	std::memcpy(&host_[offB()], b, sizeof(double) * nc_);
	return 1;
}

int32_t QPWrapperHip::updateBounds(const double lb[], const double ub[])
{
	if (!host_ && setup() != 0) return 1;
	if (lb != nullptr) std::memcpy(&host_[offLb()], lb, sizeof(double) * nv_);
	if (ub != nullptr) std::memcpy(&host_[offUb()], ub, sizeof(double) * nv_);
	return 1;
}

int32_t QPWrapperHip::solve(void)
{
	if (!dev_) {
		error_ = ASIF_HIP_ENODEVICE;
		return STATUS_UNSOLVED;
	}
	hipStream_t s = (hipStream_t)stream_;
	hipError_t e = hipSetDevice(device_);
	if (e != hipSuccess) {
		error_ = (int)e;
		return STATUS_UNSOLVED;
	}
	// batch of one: component k of the single instance sits at base[k] (ld = 1)
	int32_t *dst = (int32_t *)(dev_ + offStatus());
	const bool diag = diagonalCost_;
	const int r = asif_hip_qp_solve_batch_warm(device_, &settings, 1, 1, (int32_t)nv_, (int32_t)nc_,
	                                           diag ? dev_ + offH() : nullptr, diag ? nullptr : dev_ + offH(),
	                                           dev_ + offC(), dev_ + offA(), dev_ + offB(), dev_ + offLb(),
	                                           dev_ + offUb(), be8_.data(), dev_ + offSol(), dst, dst + 1,
	                                           dev_ + offWarmX(), dev_ + offWarmY(), warmStart && haveWarm_ ? 1 : 0,
	                                           stream_);
	if (r != 0) {
		error_ = r;
		return STATUS_UNSOLVED;
	}
	e = hipStreamSynchronize(s);
	if (e != hipSuccess) {
		error_ = (int)e;
		return STATUS_UNSOLVED;
	}
	int32_t st[2];
	std::memcpy(st, host_ + offStatus(), sizeof(st));
	status_ = st[0];
	iters_ = st[1];
	error_ = 0;
	haveWarm_ = true; // (zeros after a verdict other than "solved": the next start is cold then, as OSQP's is)
	return status_;
}

int32_t QPWrapperHip::getSolution(double sol[])
{
	if (host_)
		for (uint32_t i = 0; i < nv_; i++) sol[i] = host_[offSol() + i];
	return 1;
}

} // namespace ASIF
