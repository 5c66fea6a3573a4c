// asif_learning.h -- the learned-residual data the backup-trajectory classes carry as the public member
// `learning_data_` (reference: struct LearningData and update_weights(), include/asif_learning_utils.h:8-32,
// 123-155).  Field names are the reference's so that user code filling the struct compiles unchanged.
// Two networks (drift, actuation), each two ReLU layers and a linear one; dense column-major weights
// [rows x cols]; input [x; Dh[0..nx)] zero-padded to d_*_in; the outputs are ADDED to Lfh[0] and Lgh[0..nu).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace ASIF {

struct LearningData {
	uint32_t d_drift_in = 0, d_act_in = 0;
	uint32_t d_drift_hidden = 0, d_act_hidden = 0;
	uint32_t d_drift_hidden_2 = 0, d_act_hidden_2 = 0;
	uint32_t d_drift_out = 0, d_act_out = 0;
	const double *w_1_drift = nullptr, *w_2_drift = nullptr, *w_3_drift = nullptr;
	const double *b_1_drift = nullptr, *b_2_drift = nullptr, *b_3_drift = nullptr;
	const double *w_1_act = nullptr, *w_2_act = nullptr, *w_3_act = nullptr;
	const double *b_1_act = nullptr, *b_2_act = nullptr, *b_3_act = nullptr;
	double *Lfh_diff = nullptr; // last residuals (allocated on first use here; the reference re-allocates per call)
	double *Lgh_diff = nullptr;
};

namespace learning_detail {
// y = W v + b with W column-major [rows x cols], optionally through max(0, .)
inline void dense(const double *W, const double *b, uint32_t rows, uint32_t cols, const std::vector<double> &v,
                  bool relu, std::vector<double> &y)
{
	y.assign(rows, 0.0);
	for (uint32_t r = 0; r < rows; r++) {
		double acc = 0.0;
		for (uint32_t c = 0; c < cols; c++) acc += W[r + c * rows] * v[c];
		acc += b[r];
		y[r] = relu ? std::fmax(0., acc) : acc;
	}
}
inline void network(const double *w1, const double *b1, const double *w2, const double *b2, const double *w3,
                    const double *b3, uint32_t din, uint32_t h1, uint32_t h2, uint32_t dout,
                    const std::vector<double> &in, std::vector<double> &out)
{
	std::vector<double> a1, a2;
	dense(w1, b1, h1, din, in, true, a1);
	dense(w2, b2, h2, h1, a1, true, a2);
	dense(w3, b3, dout, h2, a2, false, out);
}
} // namespace learning_detail

inline void update_weights(LearningData *data_, const double *x, const uint32_t nx, const double *Dh, double *Lfh,
                           double *Lgh, const uint32_t nu)
{
	std::vector<double> din(data_->d_drift_in, 0.0), ain(data_->d_act_in, 0.0), dout, aout;
	for (uint32_t i = 0; i < nx; i++) {
		din[i] = x[i];
		din[i + nx] = Dh[i];
		ain[i] = x[i];
		ain[i + nx] = Dh[i];
	}
	learning_detail::network(data_->w_1_drift, data_->b_1_drift, data_->w_2_drift, data_->b_2_drift, data_->w_3_drift,
	                         data_->b_3_drift, data_->d_drift_in, data_->d_drift_hidden, data_->d_drift_hidden_2,
	                         data_->d_drift_out, din, dout);
	learning_detail::network(data_->w_1_act, data_->b_1_act, data_->w_2_act, data_->b_2_act, data_->w_3_act,
	                         data_->b_3_act, data_->d_act_in, data_->d_act_hidden, data_->d_act_hidden_2,
	                         data_->d_act_out, ain, aout);
	if (!data_->Lfh_diff) data_->Lfh_diff = new double(0.0);
	if (!data_->Lgh_diff) data_->Lgh_diff = new double[nu]();
	*data_->Lfh_diff = dout[0];
	Lfh[0] += dout[0];
	for (uint32_t i = 0; i < nu; i++) {
		Lgh[i] += aout[i];
		data_->Lgh_diff[i] = aout[i];
	}
}

} // namespace ASIF
