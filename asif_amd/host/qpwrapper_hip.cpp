// qpwrapper_hip.cpp -- see include/qpwrapper_hip.h.  Host glue only: staging buffers + one C-ABI call.
#include "qpwrapper_hip.h"
#include <hip/hip_runtime_api.h>
#include <cstring>

namespace ASIF {

QPWrapperHip::QPWrapperHip(const uint32_t nv, const uint32_t nc, const bool diagonalCost, int device)
    : QPWrapperAbstract(nv, nc, diagonalCost), device_(device), dev_(nullptr), devStatus_(nullptr), status_(0),
      iters_(0), dirty_(true)
{
	asif_hip_default_solver(&settings);
	host_.assign(total(), 0.0);
	be8_.assign(nc_ > 0 ? nc_ : 1, 0);
	sol_.assign(nv_, 0.0);
}

QPWrapperHip::~QPWrapperHip(void)
{
	if (dev_) (void)hipFree(dev_);
	if (devStatus_) (void)hipFree(devStatus_);
}

int32_t QPWrapperHip::initialize(const double H[], const double c[], const double A[], const double b[],
                                 const double lb[], const double ub[], const bool be[])
{
	if (!diagonalCost_) return ASIF_HIP_EUNSUPPORTED; // every shipped configuration uses a diagonal cost
	if (be != nullptr)
		for (uint32_t i = 0; i < nc_; i++) be_[i] = be[i];
	for (uint32_t i = 0; i < nc_; i++) be8_[i] = be_[i] ? 1 : 0;
	updateCost(H, c);
	updateA(A);
	updateb(b);
	updateBounds(lb, ub);
	if (hipSetDevice(device_) != hipSuccess) return ASIF_HIP_ENODEVICE;
	if (!dev_ && hipMalloc((void **)&dev_, sizeof(double) * total()) != hipSuccess) return ASIF_HIP_ENODEVICE;
	if (!devStatus_ && hipMalloc((void **)&devStatus_, sizeof(int32_t) * 2) != hipSuccess) return ASIF_HIP_ENODEVICE;
	// a first solve validates the shape against the compiled kernels (the reference also solves once
	// at the end of every initialize(), e.g. src/asif.cpp:101-102)
	const int32_t r = solve();
	return (r == ASIF_HIP_EUNSUPPORTED || r == ASIF_HIP_ENODEVICE || r == ASIF_HIP_EINVAL) ? r : 0;
}

int32_t QPWrapperHip::updateCost(const double H[], const double c[])
{
	if (H != nullptr)
		for (uint32_t i = 0; i < nv_; i++) host_[offHd() + i] = H[i + i * nv_]; // diagonal only, :267-272
	if (c != nullptr) std::memcpy(&host_[offC()], c, sizeof(double) * nv_);
	dirty_ = true;
	return 1;
}

int32_t QPWrapperHip::updateA(const double A[])
{
	std::memcpy(&host_[offA()], A, sizeof(double) * nc_ * nv_);
	dirty_ = true;
	return 1;
}

int32_t QPWrapperHip::updateb(const double b[])
{
	std::memcpy(&host_[offB()], b, sizeof(double) * nc_);
	dirty_ = true;
	return 1;
}

int32_t QPWrapperHip::updateBounds(const double lb[], const double ub[])
{
	if (lb != nullptr) std::memcpy(&host_[offLb()], lb, sizeof(double) * nv_);
	if (ub != nullptr) std::memcpy(&host_[offUb()], ub, sizeof(double) * nv_);
	dirty_ = true;
	return 1;
}

int32_t QPWrapperHip::solve(void)
{
	if (!dev_) return ASIF_HIP_ENODEVICE;
	if (hipSetDevice(device_) != hipSuccess) return ASIF_HIP_ENODEVICE;
	if (hipMemcpy(dev_, host_.data(), sizeof(double) * offSol(), hipMemcpyHostToDevice) != hipSuccess)
		return ASIF_HIP_ENODEVICE;
	// batch of one: component k of the single instance sits at base[k] (ld = 1)
	const int r = asif_hip_qp_solve_batch(device_, &settings, 1, 1, (int32_t)nv_, (int32_t)nc_, dev_ + offHd(),
	                                      dev_ + offC(), dev_ + offA(), dev_ + offB(), dev_ + offLb(), dev_ + offUb(),
	                                      be8_.data(), dev_ + offSol(), devStatus_, devStatus_ + 1, nullptr);
	if (r != 0) return r;
	int32_t st[2];
	if (hipMemcpy(st, devStatus_, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return ASIF_HIP_ENODEVICE;
	if (hipMemcpy(sol_.data(), dev_ + offSol(), sizeof(double) * nv_, hipMemcpyDeviceToHost) != hipSuccess)
		return ASIF_HIP_ENODEVICE;
	status_ = st[0];
	iters_ = st[1];
	dirty_ = false;
	return status_;
}

int32_t QPWrapperHip::getSolution(double sol[])
{
	for (uint32_t i = 0; i < nv_; i++) sol[i] = sol_[i];
	return 1;
}

} // namespace ASIF
