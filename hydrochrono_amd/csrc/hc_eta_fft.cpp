// hc_eta_fft.cpp -- free-surface table eta(t_j) = sum_i a_i cos(phi_i - w_i t_j) by a chirp-z transform on rocFFT.
//
// Alternative to the direct FP64 sum (eta_kernel) for the spectrum -> time-series step of IrregularWaves::
// CreateFreeSurfaceElevation (src/wave_types.cpp:717-774).  Frequencies and times are both uniform grids
// (w_i = w_0 + i*dw, t_j = t_0 + j*dt) but dw*dt*M != 2*pi in general, so a plain inverse FFT does not apply; Bluestein's
// identity  i*j = (i^2 + j^2 - (j-i)^2)/2  turns the sum into a convolution with a chirp, evaluated with three FFTs of
// length M >= nt + nf - 1:
//     sum_i c_i e^{-i w_i t_j} = e^{-i(w_0 t_0 + w_0 j dt)} e^{-i a j^2/2} * sum_i [c_i e^{-i i dw t_0} e^{-i a i^2/2}] h(j-i),
//     h(n) = e^{+i a n^2/2},  a = dw*dt,  c_i = a_i e^{i phi_i}.
// The chirp phases reach ~1e5 rad, so they are formed on the host in long double (64-bit mantissa) and reduced there;
// the device does the FFTs, the pointwise product and the final real part.  Agreement with the direct sum: ~1e-12 of max|eta|.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cmath>
#include <complex>
#include <stdexcept>
#include <string>
#include <vector>

#include "hc_context.hpp"

namespace hc {

namespace {

__global__ void __launch_bounds__(256) cmul_kernel(double2* __restrict__ a, const double2* __restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 x = a[i], y = b[i];
    a[i] = make_double2(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x);
}

// eta[j] = Re(post[j] * g[j]) / M, then the ramp rule of src/wave_types.cpp:759-769
__global__ void __launch_bounds__(256) eta_from_chirp_kernel(const double2* __restrict__ g, const double2* __restrict__ post,
                                                             const double* __restrict__ t, int nt, double inv_m, double ramp,
                                                             double* __restrict__ eta) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nt) return;
    double v = (post[j].x * g[j].x - post[j].y * g[j].y) * inv_m;
    const double tj = t[j];
    if (ramp > 0.0 && tj < ramp) {
        if (tj <= 0.0) v *= 0.0;
        else v *= tj / ramp;
    }
    eta[j] = v;
}

void fft_check(rocfft_status st, const char* what) {
    if (st != rocfft_status_success) throw Error(HC_ERR_DEVICE, std::string("rocFFT: ") + what + " failed");
}

struct FftPlan {
    rocfft_plan plan                = nullptr;
    rocfft_execution_info info      = nullptr;
    void* work                      = nullptr;
    FftPlan(rocfft_transform_type type, size_t n, hipStream_t stream) {
        fft_check(rocfft_plan_create(&plan, rocfft_placement_inplace, type, rocfft_precision_double, 1, &n, 1, nullptr), "plan_create");
        fft_check(rocfft_execution_info_create(&info), "execution_info_create");
        fft_check(rocfft_execution_info_set_stream(info, stream), "set_stream");
        size_t wsz = 0;
        fft_check(rocfft_plan_get_work_buffer_size(plan, &wsz), "get_work_buffer_size");
        if (wsz) {
            HC_HIP(hipMalloc(&work, wsz));
            fft_check(rocfft_execution_info_set_work_buffer(info, work, wsz), "set_work_buffer");
        }
    }
    ~FftPlan() {
        if (info) rocfft_execution_info_destroy(info);
        if (plan) rocfft_plan_destroy(plan);
        if (work) (void)hipFree(work);
    }
    void run(void* buf) {
        void* in[1] = {buf};
        fft_check(rocfft_execute(plan, in, nullptr, info), "execute");
    }
};

}  // namespace

// t, amp, omega, phase on the host; d_t / d_eta on the device (nt entries).  Requires nf >= 2, nt >= 2.
void eta_synthesis_fft(const std::vector<double>& t, const std::vector<double>& amp, const std::vector<double>& omega,
                       const std::vector<double>& phase, double ramp, const double* d_t, double* d_eta, hipStream_t stream) {
    const int nt = static_cast<int>(t.size()), nf = static_cast<int>(amp.size());
    if (nt < 2 || nf < 2) throw Error(HC_ERR_INVALID, "chirp-z eta synthesis needs at least two times and two components");
    static bool setup_done = false;
    if (!setup_done) {
        fft_check(rocfft_setup(), "setup");
        setup_done = true;
    }
    using ld = long double;
    const ld two_pi = 6.283185307179586476925286766559005768L;
    const ld w0 = omega[0], dw = (ld(omega[nf - 1]) - ld(omega[0])) / ld(nf - 1);
    const ld t0 = t[0], dt = (ld(t[nt - 1]) - ld(t[0])) / ld(nt - 1);
    const ld a  = dw * dt;
    size_t M = 1;
    while (M < static_cast<size_t>(nt + nf - 1)) M <<= 1;
    auto cis = [&](ld ph) {  // e^{i ph}, phase reduced in long double first
        ph = std::remainder(ph, two_pi);
        return std::complex<double>(static_cast<double>(std::cos(ph)), static_cast<double>(std::sin(ph)));
    };
    std::vector<std::complex<double>> y(M, 0.0), h(M, 0.0), post(nt);
    for (int i = 0; i < nf; ++i) {
        const ld ii = i;
        y[i] = amp[i] * cis(ld(phase[i]) - ii * dw * t0 - a * ii * ii / 2);
    }
    for (long n = -(nf - 1); n <= nt - 1; ++n) {
        const ld nn = n;
        h[static_cast<size_t>((n + static_cast<long>(M)) % static_cast<long>(M))] = cis(a * nn * nn / 2);
    }
    for (int j = 0; j < nt; ++j) {
        const ld jj = j;
        post[j] = cis(-(w0 * t0 + w0 * jj * dt) - a * jj * jj / 2);
    }
    DeviceBuffer<double2> d_y, d_h, d_post;
    d_y.alloc(M);
    d_h.alloc(M);
    d_post.alloc(nt);
    HC_HIP(hipMemcpyAsync(d_y.p, y.data(), M * sizeof(double2), hipMemcpyHostToDevice, stream));
    HC_HIP(hipMemcpyAsync(d_h.p, h.data(), M * sizeof(double2), hipMemcpyHostToDevice, stream));
    HC_HIP(hipMemcpyAsync(d_post.p, post.data(), nt * sizeof(double2), hipMemcpyHostToDevice, stream));
    {
        FftPlan fwd(rocfft_transform_type_complex_forward, M, stream), inv(rocfft_transform_type_complex_inverse, M, stream);
        fwd.run(d_y.p);
        fwd.run(d_h.p);
        hipLaunchKernelGGL(cmul_kernel, dim3(static_cast<unsigned>((M + 255) / 256)), dim3(256), 0, stream, d_y.p, d_h.p, static_cast<int>(M));
        inv.run(d_y.p);
        hipLaunchKernelGGL(eta_from_chirp_kernel, dim3((nt + 255) / 256), dim3(256), 0, stream, d_y.p, d_post.p, d_t, nt,
                           1.0 / static_cast<double>(M), ramp, d_eta);
        HC_HIP(hipGetLastError());
        HC_HIP(hipStreamSynchronize(stream));  // plans, work buffers and host staging vectors die here
    }
}

}  // namespace hc
