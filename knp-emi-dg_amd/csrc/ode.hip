// Batched membrane ODE integrator: one thread per membrane facet, per-thread adaptive Dormand-Prince 5(4)
// (same pair, step controller and tolerances as knpemidg/membrane.py:integrate_batch, which it replaces on
// the device).  Every row carries its own time and step, so a facet's result does not depend on which other
// facets share the launch (ranks that both hold a cut facet compute bitwise identical outputs).
// Stands in for the per-facet numbalsoda.lsoda loop (reference: src/knpemidg/membrane.py:84-119) for the
// membrane models of the idealized examples (reference: examples/idealized-geometries/mm_hh.py:118-161).
#include "../../include/knpemi_hip.h"
#include "knpemi_internal.hpp"

#define ODE_MAX_STATES 11
#define ODE_MAX_PARAMS 20
#define ODE_MAX_STIM 4
#define ODE_MAX_OPS 12

struct OdeSet {
    int model = 0;              // 1 = HH with synaptic stimulus, 2 = HH without, 3 = EMIx HH (cm/ms/mV), 4 = glial, 5 = passive leak, 6 = EMIx calibration system
    int ns = 0, np = 0;
    int64_t n = 0;
    int32_t* facet = nullptr;   // [n] facet id of every node
    double* states = nullptr;   // [n][ns]
    double* params = nullptr;   // [n][np]
    double* h = nullptr;        // [n] last accepted step size
    int* fail = nullptr;        // device flag: the context's status word KNP_ODE_FAIL_SLOT (read back with the next status poll)
    uint8_t* stim_mask = nullptr;   // [n] rows on which the stimulus parameters are re-imposed every step (membrane.py:98-104)
    int n_stim = 0;
    int stim_col[ODE_MAX_STIM] = {0};
    double stim_val[ODE_MAX_STIM] = {0};
};

struct StimArgs { int n; int col[ODE_MAX_STIM]; double val[ODE_MAX_STIM]; };

static std::map<knp_ctx*, std::vector<OdeSet>> g_ode;

// Hodgkin-Huxley squid axon + leak + Na/K pump (+ decaying synaptic conductance), SI units.
// parameter layout (mm_hh.py:56-64): 0 g_Na_bar 1 g_K_bar 2 g_leak_Na 3 g_leak_K 4 E_Na 5 E_K 6 Cm 7 stim_amplitude
//  8 I_ch_Na 9 I_ch_K 10 I_ch_Cl 11 K_e 12 Na_i 13 m_K 14 m_Na 15 I_max 16 E_Cl ; states: m h n V
template <bool STIM> __device__ __forceinline__ void hh_rhs(double t, const double* y, double* p, double* dy) {
    const double m = y[0], h = y[1], n = y[2], V = y[3];
    const double u = 1.0e3 * (V + 65.0e-3);
    const double alpha_m = 0.1e3 * (25.0 - u) / (exp((25.0 - u) / 10.0) - 1.0);
    const double beta_m = 4.0e3 * exp(-u / 18.0);
    dy[0] = (1 - m) * alpha_m - m * beta_m;
    const double alpha_h = 0.07e3 * exp(-u / 20.0);
    const double beta_h = 1.0e3 / (exp((30.0 - u) / 10.0) + 1.0);
    dy[1] = (1 - h) * alpha_h - h * beta_h;
    const double alpha_n = 0.01e3 * (10.0 - u) / (exp((10.0 - u) / 10.0) - 1.0);
    const double beta_n = 0.125e3 * exp(-u / 80.0);
    dy[2] = (1 - n) * alpha_n - n * beta_n;
    const double a = 1 + p[13] / p[11], b = 1 + p[14] / p[12];
    const double i_pump = p[15] / (a * a * b * b * b);
    double g_stim = 0.0;
    if (STIM) g_stim = (t < 125e-3) ? p[7] * exp(-fmod(t, 0.03) / 0.002) : 0.0;
    const double i_Na = (p[2] + p[0] * h * m * m * m + g_stim) * (V - p[4]) + 3 * i_pump;
    const double n2 = n * n;
    const double i_K = (p[3] + p[1] * n2 * n2) * (V - p[5]) - 2 * i_pump;
    p[8] = i_Na;
    p[9] = i_K;
    p[10] = 0.0;
    dy[3] = (-i_K - i_Na) / p[6];
}

// EMIx neuron membrane: the same HH kinetics in cm / ms / mV units with a periodically re-triggered synaptic
// conductance (reference: examples/emix-simulations/mm_hh.py:118-161); same parameter layout as above.
__device__ __forceinline__ void hh_emix_rhs(double t, const double* y, double* p, double* dy) {
    const double m = y[0], h = y[1], n = y[2], V = y[3];
    const double alpha_m = 0.1 * (V + 40.0) / (1.0 - exp(-(V + 40.0) / 10.0));
    const double beta_m = 4.0 * exp(-(V + 65.0) / 18.0);
    const double alpha_h = 0.07 * exp(-(V + 65.0) / 20.0);
    const double beta_h = 1.0 / (1.0 + exp(-(V + 35.0) / 10.0));
    const double alpha_n = 0.01 * (V + 55.0) / (1.0 - exp(-(V + 55.0) / 10.0));
    const double beta_n = 0.125 * exp(-(V + 65.0) / 80.0);
    dy[0] = (1 - m) * alpha_m - m * beta_m;
    dy[1] = (1 - h) * alpha_h - h * beta_h;
    dy[2] = (1 - n) * alpha_n - n * beta_n;
    const double g_stim = p[7] * exp(-fmod(t, 20.0) / 2.0);
    const double a = 1 + p[13] / p[11], b = 1 + p[14] / p[12];
    const double i_pump = p[15] / (a * a * b * b * b);
    const double i_Na = (p[2] + p[0] * h * m * m * m + g_stim) * (V - p[4]) + 3 * i_pump;
    const double n2 = n * n;
    const double i_K = (p[3] + p[1] * n2 * n2) * (V - p[5]) - 2 * i_pump;
    p[8] = i_Na;
    p[9] = i_K;
    p[10] = 0.0;
    dy[3] = (-i_K - i_Na) / p[6];
}

// Glial membrane: Kir 4.1 + Na leak + Na/K pump, one state (reference: examples/emix-simulations/mm_glial.py:117-170).
// parameters 0..15 as above, 16 K_e_init, 17 K_i_init, 18 E_Cl
__device__ __forceinline__ void glial_rhs(double t, const double* y, double* p, double* dy) {
    const double V = y[0];
    const double a = 1 + p[13] / p[11], b = 1 + p[14] / p[12];
    const double i_pump = p[15] / (a * a * b * b * b);
    const double E_K_init = 8.314e3 * 300e3 / 96485e3 * log(p[16] / p[17]);
    const double dphi = V - p[5];
    const double A = 1 + exp(18.4 / 42.4);
    const double B = 1 + exp(-(0.1186e3 + E_K_init) / 0.0441e3);
    const double C = 1 + exp((dphi + 0.0185e3) / 0.0425e3);
    const double D = 1 + exp(-(0.1186e3 + V) / 0.0441e3);
    const double g_Kir = sqrt(p[11] / p[16]) * (A * B) / (C * D);
    const double i_Kir = p[3] * g_Kir * (V - p[5]);
    const double i_Na = p[2] * (V - p[4]) + 3 * i_pump;
    const double i_K = i_Kir - 2 * i_pump;
    p[8] = i_Na;
    p[9] = i_K;
    p[10] = 0.0;
    dy[0] = (-i_K - i_Na) / p[6];
}

// Passive membrane: Na / K leak + Na/K pump + decaying synaptic conductance on the Na leak, one state, SI units
// (reference: examples/rat-neuron/mm_leak.py:107-133).  parameters (mm_leak.py:44-50): 0 g_leak_Na 1 g_leak_K 2 E_Na 3 E_K 4 Cm
//  5 stim_amplitude 6 I_ch_Na 7 I_ch_K 8 I_ch_Cl 9 K_e 10 Na_i 11 m_K 12 m_Na 13 I_max 14 E_Cl
__device__ __forceinline__ void leak_rhs(double t, const double* y, double* p, double* dy) {
    const double V = y[0];
    const double g_stim = p[5] * exp(-fmod(t, 0.03) / 0.002);
    const double a = 1 + p[11] / p[9], b = 1 + p[12] / p[10];
    const double i_pump = p[13] / (a * a * b * b * b);
    const double i_Na = (p[0] + g_stim) * (V - p[2]) + 3 * i_pump;
    const double i_K = p[1] * (V - p[3]) - 2 * i_pump;
    p[6] = i_Na;
    p[7] = i_K;
    p[8] = 0.0;
    dy[0] = (-i_K - i_Na) / p[4];
}

// EMIx calibration system: the neuronal HH membrane and the glial membrane above coupled through the ECS / neuron / glia compartment
// concentrations their currents change; integrated alone until stationary it yields the initial state of the full run
// (reference: examples/emix-simulations/mm_calibration.py:143-255, run_calibration.py:13-90).
// states: m h n V_n V_g K_e K_n K_g Na_e Na_n Na_g ; parameters: 0 g_Na_bar 1 g_K_bar 2 g_leak_Na_n 3 g_leak_K_n 4 g_leak_Na_g
//  5 g_leak_K_g 6 Cm 7 stim_amplitude 8 m_K 9 m_Na 10 I_max_n 11 I_max_g
__device__ __forceinline__ void calibration_rhs(double t, const double* y, double* p, double* dy) {
    const double m = y[0], h = y[1], n = y[2], Vn = y[3], Vg = y[4];
    const double K_e = y[5], K_n = y[6], K_g = y[7], Na_e = y[8], Na_n = y[9], Na_g = y[10];
    const double c = 8.314e3 * 300e3 / 96485e3;
    const double ICS_vol = 3.42e-11 / 2.0, ECS_vol = 7.08e-11, surface = 2.29e-6, F = 96485e3;
    const double K_g_init = 102.74050220804774, K_e_init = 3.32597273958481;
    const double E_Na_n = c * log(Na_e / Na_n), E_K_n = c * log(K_e / K_n);
    const double E_Na_g = c * log(Na_e / Na_g), E_K_g = c * log(K_e / K_g);
    const double E_K_init = c * log(K_e_init / K_g_init);
    const double alpha_m = 0.1 * (Vn + 40.0) / (1.0 - exp(-(Vn + 40.0) / 10.0));
    const double beta_m = 4.0 * exp(-(Vn + 65.0) / 18.0);
    const double alpha_h = 0.07 * exp(-(Vn + 65.0) / 20.0);
    const double beta_h = 1.0 / (1.0 + exp(-(Vn + 35.0) / 10.0));
    const double alpha_n = 0.01 * (Vn + 55.0) / (1.0 - exp(-(Vn + 55.0) / 10.0));
    const double beta_n = 0.125 * exp(-(Vn + 65.0) / 80.0);
    dy[0] = (1 - m) * alpha_m - m * beta_m;
    dy[1] = (1 - h) * alpha_h - h * beta_h;
    dy[2] = (1 - n) * alpha_n - n * beta_n;
    const double g_stim = p[7] * exp(-fmod(t, 20.0) / 2.0);
    const double a = 1 + p[8] / K_e, bn = 1 + p[9] / Na_n, bg = 1 + p[9] / Na_g;
    const double i_pump_n = p[10] / (a * a * bn * bn * bn);
    const double i_pump_g = p[11] / (a * a * bg * bg * bg);
    const double A = 1 + exp(18.4 / 42.4);
    const double B = 1 + exp(-(0.1186e3 + E_K_init) / 0.0441e3);
    const double C = 1 + exp((Vg - E_K_g + 0.0185e3) / 0.0425e3);
    const double D = 1 + exp(-(0.1186e3 + Vg) / 0.0441e3);
    const double i_Kir = p[5] * sqrt(K_e / K_e_init) * (A * B) / (C * D) * (Vg - E_K_g);
    const double i_Na_n = (p[2] + p[0] * h * m * m * m + g_stim) * (Vn - E_Na_n) + 3 * i_pump_n;
    const double n2 = n * n;
    const double i_K_n = (p[3] + p[1] * n2 * n2) * (Vn - E_K_n) - 2 * i_pump_n;
    const double i_Na_g = p[4] * (Vg - E_Na_g) + 3 * i_pump_g;
    const double i_K_g = i_Kir - 2 * i_pump_g;
    dy[3] = (-i_K_n - i_Na_n) / p[6];
    dy[4] = (-i_K_g - i_Na_g) / p[6];
    const double ke = surface / (F * ECS_vol), ki = surface / (F * ICS_vol);
    dy[5] = (i_K_n + i_K_g) * ke;
    dy[6] = -i_K_n * ki;
    dy[7] = -i_K_g * ki;
    dy[8] = (i_Na_n + i_Na_g) * ke;
    dy[9] = -i_Na_n * ki;
    dy[10] = -i_Na_g * ki;
}

template <int MODEL> __device__ __forceinline__ void model_rhs(double t, const double* y, double* p, double* dy) {
    if (MODEL == 1) hh_rhs<true>(t, y, p, dy);
    else if (MODEL == 2) hh_rhs<false>(t, y, p, dy);
    else if (MODEL == 3) hh_emix_rhs(t, y, p, dy);
    else if (MODEL == 4) glial_rhs(t, y, p, dy);
    else if (MODEL == 5) leak_rhs(t, y, p, dy);
    else calibration_rhs(t, y, p, dy);
}

template <int MODEL, int NS, int NP>
__global__ __launch_bounds__(64) void k_ode_step(int64_t n, double t0, double t1, double rtol, double atol, int max_steps,
                                                 double* __restrict__ states, double* __restrict__ params,
                                                 double* __restrict__ hstore, int* __restrict__ fail,
                                                 const uint8_t* __restrict__ stim_mask, StimArgs stim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double y[NS], p[NP], k[7][NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) y[s] = states[i * NS + s];
#pragma unroll
    for (int q = 0; q < NP; ++q) p[q] = params[i * NP + q];
    // the stimulus overwrites its parameters on the masked rows at the start of EVERY step (membrane.py:102-104), whatever a
    // hook or a parameter upload wrote there in between
    if (stim.n > 0 && stim_mask[i]) {
        for (int e = 0; e < stim.n; ++e)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                if (q == stim.col[e]) p[q] = stim.val[e];
    }
    // Dormand-Prince 5(4)
    const double C[7] = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1, 1};
    const double A[7][6] = {{0, 0, 0, 0, 0, 0},
                            {1.0 / 5, 0, 0, 0, 0, 0},
                            {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
                            {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
                            {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
                            {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
                            {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
    const double B5[7] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0};
    const double B4[7] = {5179.0 / 57600, 0, 7571.0 / 16695, 393.0 / 640, -92097.0 / 339200, 187.0 / 2100, 1.0 / 40};
    double t = t0;
    double h = hstore[i];
    if (!(h > 0.0)) h = (t1 - t0) / 16;
    const double tiny = 1e-14 * fmax(fabs(t1), 1e-30);
    model_rhs<MODEL>(t, y, p, k[0]);
    int steps = 0;
    bool done = false;
    while (!done && steps < max_steps) {
        const double hh = fmin(h, t1 - t);
#pragma unroll
        for (int s = 1; s < 7; ++s) {
            double ys[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    if (j < s) acc += A[s][j] * k[j][q];
                ys[q] = y[q] + hh * acc;
            }
            model_rhs<MODEL>(t + C[s] * hh, ys, p, k[s]);
        }
        double e = 0.0, y5[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            double a5 = 0.0, ae = 0.0;
#pragma unroll
            for (int j = 0; j < 7; ++j) { a5 += B5[j] * k[j][q]; ae += (B5[j] - B4[j]) * k[j][q]; }
            y5[q] = y[q] + hh * a5;
            const double scale = fmax(atol + rtol * fmax(fabs(y[q]), fabs(y5[q])), 1e-300);   // atol = 0 is the reference's (membrane.py:112)
            e = fmax(e, fabs(hh * ae) / scale);
        }
        if (!(e == e) || isinf(e)) e = 1e10;
        ++steps;
        if (e <= 1.0 || hh < tiny) {
            t += hh;
#pragma unroll
            for (int q = 0; q < NS; ++q) { y[q] = y5[q]; k[0][q] = k[6][q]; }
            if (t >= t1 - 1e-15 * fabs(t1)) done = true;
        }
        const double fac = fmin(5.0, fmax(0.2, 0.9 * pow(1.0 / fmax(e, 1e-10), 0.2)));
        if (!done) h = hh * fac;
    }
    if (!done) atomicExch(fail, 1);
    model_rhs<MODEL>(t1, y, p, k[0]);                 // leave I_ch_k evaluated at the end state
#pragma unroll
    for (int s = 0; s < NS; ++s) states[i * NS + s] = y[s];
#pragma unroll
    for (int q = 0; q < NP; ++q) params[i * NP + q] = p[q];
    hstore[i] = h;
}

// table[node][col] <- facet_field[facet[node]]   (PDE -> ODE, membrane.py:122-139)
__global__ void k_ode_from_facet(int64_t n, const int32_t* __restrict__ facet, const double* __restrict__ field, int stride,
                                 int col, double* __restrict__ table) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) table[i * stride + col] = field[facet[i]];
}
// facet_field[facet[node]] <- table[node][col]   (ODE -> PDE, membrane.py:141-162)
__global__ void k_ode_to_facet(int64_t n, const int32_t* __restrict__ facet, const double* __restrict__ table, int stride,
                               int col, double* __restrict__ field) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) field[facet[i]] = table[i * stride + col];
}

// the same for a list of columns in ONE launch (a membrane step moves V, E_k, K_e, Na_i in and V, I_ch_k out)
struct OdeOps { int n; int stride[ODE_MAX_OPS]; int col[ODE_MAX_OPS]; double* table[ODE_MAX_OPS]; double* field[ODE_MAX_OPS]; };
__global__ void k_ode_exchange_multi(int64_t n, const int32_t* __restrict__ facet, OdeOps ops, int to_facet) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = facet[i];
    for (int k = 0; k < ops.n; ++k) {
        double* t = ops.table[k] + i * ops.stride[k] + ops.col[k];
        if (to_facet) ops.field[k][f] = *t;
        else *t = ops.field[k][f];
    }
}

double* knp_field_ptr(knp_ctx* c, int field, int64_t* n);   // abi.hip

// `assert success` of membrane.py:113, deferred: knp_ode_step does not synchronise; the flag travels with the next status poll of
// a solve (krylov.hip: poll_status), a table download or knp_sync.  Call with the stream idle.
int ode_check_failed(knp_ctx* c) {
    int fail = 0;
    if (hipMemcpy(&fail, c->status + KNP_ODE_FAIL_SLOT, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (!fail) return 0;
    hipMemset(c->status + KNP_ODE_FAIL_SLOT, 0, sizeof(int));
    c->err = "ODE integrator did not reach the end time";
    return 1;
}

static OdeSet* get_set(knp_ctx* c, int handle) {
    auto it = g_ode.find(c);
    if (it == g_ode.end() || handle < 0 || handle >= (int)it->second.size()) return nullptr;
    return &it->second[handle];
}

void ode_destroy_all(knp_ctx* c) {
    auto it = g_ode.find(c);
    if (it == g_ode.end()) return;
    for (auto& S : it->second) { hipFree(S.facet); hipFree(S.states); hipFree(S.params); hipFree(S.h); hipFree(S.stim_mask); }
    g_ode.erase(it);
}

extern "C" {

int knp_ode_create(knp_ctx* c, int model, int64_t n, const int32_t* facets, int ns, int np, const double* states,
                   const double* params) {
    if (!c) return -1;
    if (model < 1 || model > 6) { c->err = "ode: unknown device model id"; return -1; }
    if (model <= 3 && (ns != 4 || np != 17)) { c->err = "ode: HH models have 4 states and 17 parameters"; return -1; }
    if (model == 4 && (ns != 1 || np != 19)) { c->err = "ode: the glial model has 1 state and 19 parameters"; return -1; }
    if (model == 5 && (ns != 1 || np != 15)) { c->err = "ode: the leak model has 1 state and 15 parameters"; return -1; }
    if (model == 6 && (ns != 11 || np != 12)) { c->err = "ode: the calibration system has 11 states and 12 parameters"; return -1; }
    for (int64_t i = 0; i < n; ++i)
        if (facets[i] < 0 || facets[i] >= c->m.nf) { c->err = "ode: facet id out of range"; return -1; }
    OdeSet S;
    S.model = model; S.ns = ns; S.np = np; S.n = n;
    const size_t m = (size_t)(n ? n : 1);
    HIPCHK(c, hipMalloc((void**)&S.facet, m * sizeof(int32_t)));
    HIPCHK(c, hipMalloc((void**)&S.states, m * ns * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&S.params, m * np * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&S.h, m * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&S.stim_mask, m));
    HIPCHK(c, hipMemset(S.stim_mask, 0, m));
    S.fail = c->status + KNP_ODE_FAIL_SLOT;
    HIPCHK(c, hipMemset(S.h, 0, m * sizeof(double)));
    if (n) {
        HIPCHK(c, hipMemcpy(S.facet, facets, n * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(S.states, states, n * ns * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(S.params, params, n * np * sizeof(double), hipMemcpyHostToDevice));
    }
    g_ode[c].push_back(S);
    return (int)g_ode[c].size() - 1;
}

// what: 0 = states, 1 = parameters
int knp_ode_table(knp_ctx* c, int handle, int what, int upload, double* host) {
    OdeSet* S = get_set(c, handle);
    if (!S || !host) return -1;
    double* dev = what == 0 ? S->states : S->params;
    const size_t bytes = (size_t)S->n * (what == 0 ? S->ns : S->np) * sizeof(double);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (ode_check_failed(c)) return -4;
    if (bytes) HIPCHK(c, hipMemcpy(upload ? (void*)dev : (void*)host, upload ? (void*)host : (void*)dev, bytes,
                                   upload ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
    return 0;
}

// ODE table column <- device facet field (offset = row * nf for multi-row fields), or the reverse
int knp_ode_exchange(knp_ctx* c, int handle, int what, int col, int field, int64_t offset, int to_facet) {
    OdeSet* S = get_set(c, handle);
    if (!S) return -1;
    int64_t nfld = 0;
    double* f = knp_field_ptr(c, field, &nfld);
    const int stride = what == 0 ? S->ns : S->np;
    if (!f || col < 0 || col >= stride || offset < 0 || offset + c->m.nf > nfld) { c->err = "ode_exchange: bad field / column"; return -1; }
    if (!S->n) return 0;
    double* table = what == 0 ? S->states : S->params;
    const dim3 g((unsigned)((S->n + 255) / 256)), b(256);
    if (to_facet) hipLaunchKernelGGL(k_ode_to_facet, g, b, 0, c->stream, S->n, S->facet, table, stride, col, f + offset);
    else hipLaunchKernelGGL(k_ode_from_facet, g, b, 0, c->stream, S->n, S->facet, f + offset, stride, col, table);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int knp_ode_set_stimulus(knp_ctx* c, int handle, int n_entries, const int32_t* cols, const double* values, const uint8_t* mask) {
    OdeSet* S = get_set(c, handle);
    if (!S) return -1;
    if (n_entries < 0 || n_entries > ODE_MAX_STIM) { c->err = "ode_set_stimulus: at most 4 stimulus parameters"; return -1; }
    for (int e = 0; e < n_entries; ++e)
        if (cols[e] < 0 || cols[e] >= S->np) { c->err = "ode_set_stimulus: parameter column out of range"; return -1; }
    S->n_stim = n_entries;
    for (int e = 0; e < n_entries; ++e) { S->stim_col[e] = cols[e]; S->stim_val[e] = values[e]; }
    if (n_entries && S->n) {
        if (!mask) { c->err = "ode_set_stimulus: mask missing"; return -1; }
        HIPCHK(c, hipMemcpyAsync(S->stim_mask, mask, (size_t)S->n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

int knp_ode_exchange_multi(knp_ctx* c, int handle, int n, const int32_t* what, const int32_t* col, const int32_t* field,
                           const int64_t* offset, int to_facet) {
    OdeSet* S = get_set(c, handle);
    if (!S || n < 0 || (n && (!what || !col || !field || !offset))) return -1;
    for (int k0 = 0; k0 < n; k0 += ODE_MAX_OPS) {
        OdeOps ops;
        ops.n = (n - k0 < ODE_MAX_OPS) ? n - k0 : ODE_MAX_OPS;
        for (int k = 0; k < ops.n; ++k) {
            int64_t nfld = 0;
            double* f = knp_field_ptr(c, field[k0 + k], &nfld);
            const int stride = what[k0 + k] == 0 ? S->ns : S->np;
            if (!f || col[k0 + k] < 0 || col[k0 + k] >= stride || offset[k0 + k] < 0 || offset[k0 + k] + c->m.nf > nfld) {
                c->err = "ode_exchange: bad field / column";
                return -1;
            }
            ops.stride[k] = stride; ops.col[k] = col[k0 + k];
            ops.table[k] = what[k0 + k] == 0 ? S->states : S->params;
            ops.field[k] = f + offset[k0 + k];
        }
        if (!S->n || !ops.n) continue;
        hipLaunchKernelGGL(k_ode_exchange_multi, dim3((unsigned)((S->n + 255) / 256)), dim3(256), 0, c->stream, S->n, S->facet, ops, to_facet);
        HIPCHK(c, hipGetLastError());
    }
    return 0;
}

// Asynchronous: a node that cannot reach t0 + dt raises the context's ODE failure flag, reported (-4) by the next solve's status
// poll, table download or knp_sync -- no host round trip per membrane model and step.
int knp_ode_step(knp_ctx* c, int handle, double t0, double dt, double rtol, double atol) {
    OdeSet* S = get_set(c, handle);
    if (!S) return -1;
    if (!S->n) return 0;
    const dim3 g((unsigned)((S->n + 63) / 64)), b(64);
    const int max_steps = 100000;
    StimArgs st;
    st.n = S->n_stim;
    for (int e = 0; e < ODE_MAX_STIM; ++e) { st.col[e] = S->stim_col[e]; st.val[e] = S->stim_val[e]; }
#define ODE_LAUNCH(MODEL, NS, NP)                                                                                          \
    hipLaunchKernelGGL((k_ode_step<MODEL, NS, NP>), g, b, 0, c->stream, S->n, t0, t0 + dt, rtol, atol, max_steps, S->states, \
                       S->params, S->h, S->fail, (const uint8_t*)S->stim_mask, st)
    if (S->model == 1) ODE_LAUNCH(1, 4, 17);
    else if (S->model == 2) ODE_LAUNCH(2, 4, 17);
    else if (S->model == 3) ODE_LAUNCH(3, 4, 17);
    else if (S->model == 4) ODE_LAUNCH(4, 1, 19);
    else if (S->model == 5) ODE_LAUNCH(5, 1, 15);
    else ODE_LAUNCH(6, 11, 12);
#undef ODE_LAUNCH
    HIPCHK(c, hipGetLastError());
    return 0;
}

}  // extern "C"
