/*
 * t1d_oracle.c -- CPU restatement of the simglucose hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (simglucose_amd/) never does.  Plain C, fp64 throughout,
 * one function per reference function, each citing the reference file:line it follows
 * (paths relative to /root/reference).  Nothing here is copied from the reference: the
 * reference is Python/NumPy/SciPy, this is a from-scratch C statement of the same
 * arithmetic, pinned against golden vectors produced by running the reference itself
 * (oracle/gen_golden.py -> tests/golden/, checked by tests/test_oracle_golden.py).
 *
 * Third-party algorithm restated here: scipy.integrate.ode('dopri5') = Hairer & Wanner's
 * DOPRI5 (Dormand-Prince 5(4), scipy pins ==1.6.3 in the reference's Pipfile.lock; the
 * installed 1.15.3 wraps the same Fortran driver).  Call sites: patient/t1dpatient.py:276-277
 * (construction, default tolerances rtol 1e-6 / atol 1e-12) and :110-113 (one integrate()
 * per simulated minute).  The step-size controller, the initial-step probe and the
 * carry-over of the predicted step between calls are restated in o_dopri5_minute().
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <float.h>

#include "t1d_oracle.h"

/* ------------------------------------------------------------------------------------------
 * P1: T1DPatient.model  (patient/t1dpatient.py:119-208)
 * x[13] state, cho = grams eaten this minute (g/min), ins = U/min, lq = last_Qsto (mg),
 * lf = last_foodtaken (g).  p points at one row of the patient table (T1D_O_* columns).
 * ---------------------------------------------------------------------------------------- */
void t1d_o_rhs(const double* p, const double* x, double cho, double ins, double lq, double lf,
               double* dx)
{
    const double d = cho * 1000.0;                       /* :121  g -> mg               */
    const double insulin = ins * 6000.0 / p[T1D_O_BW];   /* :122  U/min -> pmol/kg/min  */
    const double qsto = x[0] + x[1];                     /* :126 */
    const double Dbar = lq + lf * 1000.0;                /* :130 */
    double kgut;

    dx[0] = -p[T1D_O_KMAX] * x[0] + d;                   /* :133 */
    if (Dbar > 0.0) {                                    /* :135-140 */
        const double aa = 5.0 / 2.0 / (1.0 - p[T1D_O_B]) / Dbar;
        const double cc = 5.0 / 2.0 / p[T1D_O_D] / Dbar;
        kgut = p[T1D_O_KMIN] + (p[T1D_O_KMAX] - p[T1D_O_KMIN]) / 2.0 *
               (tanh(aa * (qsto - p[T1D_O_B] * Dbar)) - tanh(cc * (qsto - p[T1D_O_D] * Dbar)) + 2.0);
    } else {
        kgut = p[T1D_O_KMAX];                            /* :142 */
    }
    dx[1] = p[T1D_O_KMAX] * x[0] - x[1] * kgut;          /* :145 */
    dx[2] = kgut * x[1] - p[T1D_O_KABS] * x[2];          /* :148 */

    const double Rat = p[T1D_O_F] * p[T1D_O_KABS] * x[2] / p[T1D_O_BW];          /* :151 */
    const double EGPt = p[T1D_O_KP1] - p[T1D_O_KP2] * x[3] - p[T1D_O_KP3] * x[8];/* :153 */
    const double Uiit = p[T1D_O_FSNC];                                           /* :155 */
    const double Et = (x[3] > p[T1D_O_KE2]) ? p[T1D_O_KE1] * (x[3] - p[T1D_O_KE2]) : 0.0; /* :158-161 */

    dx[3] = (EGPt > 0.0 ? EGPt : 0.0) + Rat - Uiit - Et - p[T1D_O_K1] * x[3] + p[T1D_O_K2] * x[4]; /* :165 */
    dx[3] = (x[3] >= 0.0) ? dx[3] : 0.0 * dx[3];         /* :167  (bool * value)         */

    const double Vmt = p[T1D_O_VM0] + p[T1D_O_VMX] * x[6];                       /* :169 */
    const double Uidt = Vmt * x[4] / (p[T1D_O_KM0] + x[4]);                      /* :171 */
    dx[4] = -Uidt + p[T1D_O_K1] * x[3] - p[T1D_O_K2] * x[4];                     /* :172 */
    dx[4] = (x[4] >= 0.0) ? dx[4] : 0.0 * dx[4];                                 /* :173 */

    dx[5] = -(p[T1D_O_M2] + p[T1D_O_M4]) * x[5] + p[T1D_O_M1] * x[9] + p[T1D_O_KA1] * x[10] +
            p[T1D_O_KA2] * x[11];                                                /* :176 */
    const double It = x[5] / p[T1D_O_VI];                                        /* :178 */
    dx[5] = (x[5] >= 0.0) ? dx[5] : 0.0 * dx[5];                                 /* :179 */

    dx[6] = -p[T1D_O_P2U] * x[6] + p[T1D_O_P2U] * (It - p[T1D_O_IB]);            /* :182 */
    dx[7] = -p[T1D_O_KI] * (x[7] - It);                                          /* :185 */
    dx[8] = -p[T1D_O_KI] * (x[8] - x[7]);                                        /* :187 */

    dx[9] = -(p[T1D_O_M1] + p[T1D_O_M30]) * x[9] + p[T1D_O_M2] * x[5];           /* :190 */
    dx[9] = (x[9] >= 0.0) ? dx[9] : 0.0 * dx[9];                                 /* :191 */

    dx[10] = insulin - (p[T1D_O_KA1] + p[T1D_O_KD]) * x[10];                     /* :194 */
    dx[10] = (x[10] >= 0.0) ? dx[10] : 0.0 * dx[10];                             /* :195 */
    dx[11] = p[T1D_O_KD] * x[10] - p[T1D_O_KA2] * x[11];                         /* :197 */
    dx[11] = (x[11] >= 0.0) ? dx[11] : 0.0 * dx[11];                             /* :198 */
    dx[12] = (-p[T1D_O_KSC] * x[12] + p[T1D_O_KSC] * x[3]);                      /* :201 */
    dx[12] = (x[12] >= 0.0) ? dx[12] : 0.0 * dx[12];                             /* :202 */
}

/* ------------------------------------------------------------------------------------------
 * Fixed-step classical RK4 over one minute in n_sub sub-steps: the integrator the HIP kernel
 * uses in place of P2 (BASELINE.json north_star).  Inputs are constant over the minute, as
 * they are in patient/t1dpatient.py:110-113 (set_f_params once per integrate()).
 * ---------------------------------------------------------------------------------------- */
void t1d_o_rk4_minute(const double* p, double* x, double cho, double ins, double lq, double lf,
                      int n_sub)
{
    const double h = 1.0 / (double)n_sub;
    double k1[13], k2[13], k3[13], k4[13], y[13];
    for (int s = 0; s < n_sub; ++s) {
        t1d_o_rhs(p, x, cho, ins, lq, lf, k1);
        for (int i = 0; i < 13; ++i) y[i] = x[i] + 0.5 * h * k1[i];
        t1d_o_rhs(p, y, cho, ins, lq, lf, k2);
        for (int i = 0; i < 13; ++i) y[i] = x[i] + 0.5 * h * k2[i];
        t1d_o_rhs(p, y, cho, ins, lq, lf, k3);
        for (int i = 0; i < 13; ++i) y[i] = x[i] + h * k3[i];
        t1d_o_rhs(p, y, cho, ins, lq, lf, k4);
        for (int i = 0; i < 13; ++i) x[i] = x[i] + h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
}

/* ------------------------------------------------------------------------------------------
 * Multirate RK4 over one minute (integrator 2): the gastro-intestinal tract (x0, x1, x2) -- the only part
 * of the model with fast dynamics (kabs up to 1.23/min) and the tanh gastric-emptying term -- is an
 * autonomous sub-system and is integrated with classical RK4 at ng sub-steps per minute; the ten
 * glucose/insulin states, whose rates are all <= ~0.45/min, follow with classical RK4 at ns sub-steps,
 * reading the gut content x2 (rate of appearance, t1dpatient.py:151) from the fine solution at their
 * stage times t, t+H/2, t+H (ng must be a multiple of 2 ns).  Same RHS arithmetic as t1d_o_rhs.
 * ---------------------------------------------------------------------------------------- */
static void o_rhs_gut(const double* p, const double* g, double cho, double lq, double lf, double* dg)
{
    double x[13] = {0}, dx[13];
    x[0] = g[0]; x[1] = g[1]; x[2] = g[2];
    t1d_o_rhs(p, x, cho, 0.0, lq, lf, dx);
    dg[0] = dx[0]; dg[1] = dx[1]; dg[2] = dx[2];
}

void t1d_o_mr_minute(const double* p, double* x, double cho, double ins, double lq, double lf, int ng, int ns)
{
    double x2s[257];                     /* x2 on the fine grid, ng <= 256 */
    double g[3] = {x[0], x[1], x[2]};
    const double h = 1.0 / (double)ng;
    x2s[0] = g[2];
    for (int s = 0; s < ng; ++s) {
        double k1[3], k2[3], k3[3], k4[3], y[3];
        o_rhs_gut(p, g, cho, lq, lf, k1);
        for (int i = 0; i < 3; ++i) y[i] = g[i] + 0.5 * h * k1[i];
        o_rhs_gut(p, y, cho, lq, lf, k2);
        for (int i = 0; i < 3; ++i) y[i] = g[i] + 0.5 * h * k2[i];
        o_rhs_gut(p, y, cho, lq, lf, k3);
        for (int i = 0; i < 3; ++i) y[i] = g[i] + h * k3[i];
        o_rhs_gut(p, y, cho, lq, lf, k4);
        for (int i = 0; i < 3; ++i) g[i] = g[i] + h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        x2s[s + 1] = g[2];
    }
    const double H = 1.0 / (double)ns;
    const int stride = ng / ns;          /* fine steps per slow step (even) */
    for (int s = 0; s < ns; ++s) {
        double k1[13], k2[13], k3[13], k4[13], y[13];
        const double xa = x2s[s * stride], xm = x2s[s * stride + stride / 2], xb = x2s[(s + 1) * stride];
        x[2] = xa;
        t1d_o_rhs(p, x, cho, ins, lq, lf, k1);
        for (int i = 3; i < 13; ++i) y[i] = x[i] + 0.5 * H * k1[i];
        y[0] = x[0]; y[1] = x[1]; y[2] = xm;
        t1d_o_rhs(p, y, cho, ins, lq, lf, k2);
        for (int i = 3; i < 13; ++i) y[i] = x[i] + 0.5 * H * k2[i];
        t1d_o_rhs(p, y, cho, ins, lq, lf, k3);
        for (int i = 3; i < 13; ++i) y[i] = x[i] + H * k3[i];
        y[2] = xb;
        t1d_o_rhs(p, y, cho, ins, lq, lf, k4);
        for (int i = 3; i < 13; ++i) x[i] = x[i] + H / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    x[0] = g[0]; x[1] = g[1]; x[2] = g[2];
}

/* ------------------------------------------------------------------------------------------
 * Split scheme over one minute (integrators 3 and 4) -- what the HIP kernels integrate with.
 * Same model (t1dpatient.py:119-208), partitioned by what each part of it needs:
 *   insulin  s = (x5, x9, x10, x11, x6, x7, x8): linear with the minute's constant infusion (:176-198), so it
 *            is advanced with its exact propagator s(tau) = Phi(tau) [s; u; 1] (Phi from the host, one 7x9
 *            block per tau = k/(2 n_sub)); the (x >= 0) factors of :179,191,195,198 never switch on this
 *            non-negative linear flow and are dropped;
 *   gut      x0, x1 (:133-145): classical RK4; F = kgut*x1 is also integrated (Q, RK4 quadrature);
 *            x2 (:148) is linear in itself with rate kabs (up to 1.23/min): exponential form
 *            x2' = E x2 + wa F1 + wm (F2+F3)/2 + wb F4 (ETD-RK4 weights per step size from the host);
 *            absorbed mass R = x2(0) - x2 + Q (what has left x2 through kabs since the minute began);
 *   glucose  x3, x4, x12 (:151-173,201-202): classical RK4 at half as many steps as the gut, on (z3 = x3 - c R, x4, x12),
 *            c = f/BW: the fast rate-of-appearance forcing then enters only through the argument
 *            x3 = z3 + c R(tau), with R, X = x6 and XL = x8 taken at the stage times from the parts above.
 *            The (x >= 0) factors of :167,173,202 are kept.
 *            A step that begins with x3 < 0 holds x3 exactly (every stage of the reference's RHS returns dx3 = 0
 *            there, :167); a step that takes x3 from >= 0 to < 0 ends at x3 = -1e-10: scipy's step-size control
 *            (atol 1e-12) lands within ~1e-10 below zero, where the state then stays for good, and a fixed step
 *            would overshoot by up to ~0.1 mg/kg (0.05 mg/dL held for the rest of the episode).
 * Step sizes (integrator 3, "split": level 1 in every minute; integrator 4, "split_adaptive": per minute and env,
 * from the state and the rates at the start of the minute -- o_tier_level below; a deterministic rule, so the HIP
 * kernels take the same decisions):
 *   level 1  gut n_sub steps,   glucose n_sub/2   (n_sub = 4: RK4 at h = 1/4 and H = 1/2)
 *   level 2  gut 2 n_sub steps, glucose n_sub     an argument of the gastric-emptying tanh pair (:138-140) moves fast
 *            through its transition, x3 is about to reach 0 (:167), or insulin action makes the tissue compartment
 *            fast (:169-172): ~0.7 % of the env-minutes of RandomScenario days
 * Against a tight solve on random env-days: max 9.2e-4 mg/dL (level 1 everywhere: 6.9e-3); patients driven through
 * BG = 0 stay within 7e-4 of the SciPy solution -- tools/tier_study.py.  (A third, coarser level for calm minutes and a
 * finer level 2 were built and measured in round 2: 1.7e-4, but every kernel that takes its levels in place pays for
 * the most expensive lane of each wave, and the one-minute kernel is bound by HBM, not arithmetic -- DESIGN.md.)
 * ---------------------------------------------------------------------------------------- */
static double o_kgut(const double* p, double qsto, double Dbar)
{
    if (Dbar > 0.0) {
        const double aa = 5.0 / 2.0 / (1.0 - p[T1D_O_B]) / Dbar;
        const double cc = 5.0 / 2.0 / p[T1D_O_D] / Dbar;
        return p[T1D_O_KMIN] + (p[T1D_O_KMAX] - p[T1D_O_KMIN]) / 2.0 *
               (tanh(aa * (qsto - p[T1D_O_B] * Dbar)) - tanh(cc * (qsto - p[T1D_O_D] * Dbar)) + 2.0);
    }
    return p[T1D_O_KMAX];
}

static void o_glucose_rhs(const double* p, const double* y, double cR, double cRdot, double X, double XL,
                          double x3_frozen, double* dy)
{
    /* x3_frozen < 0: the step began with x3 < 0, where the reference holds it (dx3 = 0 at every stage, :167) */
    const double x3 = x3_frozen < 0.0 ? x3_frozen : y[0] + cR, x4 = y[1], x12 = y[2];
    const double EGPt = p[T1D_O_KP1] - p[T1D_O_KP2] * x3 - p[T1D_O_KP3] * XL;
    const double Et = (x3 > p[T1D_O_KE2]) ? p[T1D_O_KE1] * (x3 - p[T1D_O_KE2]) : 0.0;
    double d3 = (EGPt > 0.0 ? EGPt : 0.0) - p[T1D_O_FSNC] - Et - p[T1D_O_K1] * x3 + p[T1D_O_K2] * x4;
    d3 = (x3 >= 0.0) ? d3 : -cRdot;                      /* :167: dx3 = 0  <=>  dz3 = -c R'            */
    const double Vmt = p[T1D_O_VM0] + p[T1D_O_VMX] * X;
    double d4 = -Vmt * x4 / (p[T1D_O_KM0] + x4) + p[T1D_O_K1] * x3 - p[T1D_O_K2] * x4;
    d4 = (x4 >= 0.0) ? d4 : 0.0;                         /* :173 */
    double d12 = -p[T1D_O_KSC] * x12 + p[T1D_O_KSC] * x3;
    d12 = (x12 >= 0.0) ? d12 : 0.0;                      /* :202 */
    dy[0] = d3; dy[1] = d4; dy[2] = d12;
}

/* Knobs of the step-size rule (tools may override them through t1d_o_set_knob for studies; the defaults are what the
 * HIP kernels hard-code):
 *   0 NEAR   |tanh argument| below this somewhere in the minute ...
 *   1 MOVE   ... while it changes by more than this over the minute                      -> level 2 (gut)
 *   2 KINK   x3 >= 0 with min(|x3|, |x3 + dx3|) < KINK |dx3|, or x3 + dx3 < 0              -> level 2 (x3 reaches 0, :167)
 *   3 SNAP   x3 that a glucose step takes from >= 0 to < 0 is set to -SNAP (0 = leave it)
 *   4 STIFF  rate of the tissue compartment Vmt / (Km0 + x4) + k2 above this (1/min)      -> level 2               */
static double o_knob[5] = {3.0, 4.0, 1.0, 1e-10, 2.0};
void t1d_o_set_knob(int k, double v) { if (k >= 0 && k < 5) o_knob[k] = v; }
double t1d_o_get_knob(int k) { return (k >= 0 && k < 5) ? o_knob[k] : 0.0; }

/* The step-size rule: level of the minute from the state and the rates at its start.  dq = d(qsto)/dt, dx3 = dx3/dt
 * (rate of appearance included). */
static int o_tier_level(const double* p, const double* x, double Dbar, double dq, double dx3)
{
    int lvl2 = 0;
    if (Dbar > 0.0) {
        const double aa = 5.0 / 2.0 / (1.0 - p[T1D_O_B]) / Dbar, cc = 5.0 / 2.0 / p[T1D_O_D] / Dbar;
        const double q0 = x[0] + x[1];
        const double A0 = aa * (q0 - p[T1D_O_B] * Dbar), A1 = A0 + aa * dq;
        const double C0 = cc * (q0 - p[T1D_O_D] * Dbar), C1 = C0 + cc * dq;
        const int fa = fabs(A1 - A0) > o_knob[1] && (A0 * A1 <= 0.0 || fmin(fabs(A0), fabs(A1)) < o_knob[0]);
        const int fc = fabs(C1 - C0) > o_knob[1] && (C0 * C1 <= 0.0 || fmin(fabs(C0), fabs(C1)) < o_knob[0]);
        lvl2 = fa || fc;
    }
    const double x3 = x[3];
    if (x3 >= 0.0) {                                                     /* (x3 < 0 is held: nothing ahead) */
        const double x3e = x3 + dx3;
        if (x3e <= 0.0 || fmin(x3, x3e) < o_knob[2] * fabs(dx3)) lvl2 = 1;
    }
    /* insulin-dependent utilisation (:169-172) makes the tissue compartment fast under large insulin action */
    const double lam4 = (p[T1D_O_VM0] + p[T1D_O_VMX] * x[6]) / (p[T1D_O_KM0] + x[4]) + p[T1D_O_K2];
    if (lam4 > o_knob[4]) lvl2 = 1;
    return lvl2 ? 2 : 1;
}

/* tab: [2 n_sub][7][9] Phi(k / (2 n_sub)), k = 1 .. 2 n_sub, then (E, wa, wm, wb) for the gut step of level 1
 * (h = 1/n_sub) and of level 2 (h/2).
 * mode 0: level 1 in every minute (the fixed-step "split" scheme); mode 1: level by o_tier_level.
 * Returns the level used, or -1 on bad arguments. */
int t1d_o_split_minute(const double* p, const double* tab, double* x, double cho, double ins, double lq,
                       double lf, int n_sub, int mode)
{
    if (n_sub < 2 || n_sub > 8 || (n_sub & 1) || !tab) return -1;
    const int nb = 2 * n_sub;
    const double d = cho * 1000.0, u = ins * 6000.0 / p[T1D_O_BW], Dbar = lq + lf * 1000.0;
    const double kmax = p[T1D_O_KMAX], kabs = p[T1D_O_KABS], c = p[T1D_O_F] / p[T1D_O_BW];
    static const int SI[7] = {5, 9, 10, 11, 6, 7, 8};
    double y[3] = {x[3], x[4], x[12]}, k1[3], k2[3], k3[3], k4[3], w[3];
    int level = 1;
    if (mode == 1) {
        const double F1 = o_kgut(p, x[0] + x[1], Dbar) * x[1];
        double k0[3];
        o_glucose_rhs(p, y, 0.0, 0.0, x[6], x[8], 0.0, k0);           /* dz3 without the Rat term ... */
        const double dx3 = k0[0] + ((x[3] >= 0.0) ? c * kabs * x[2] : 0.0);   /* ... so dx3 = dz3 + c kabs x2 (:151,165) */
        level = o_tier_level(p, x, Dbar, d - F1, dx3);
    }
    const int nh = level == 1 ? n_sub : 2 * n_sub;                        /* gut steps = glucose half steps */
    const int ns = nh / 2;                                                /* glucose steps */
    const int sb = nb / nh;                                               /* table blocks per half step */
    /* insulin: exact propagation to every tau = k/nh */
    double aug[9], S[17][7];
    for (int j = 0; j < 7; ++j) { aug[j] = x[SI[j]]; S[0][j] = aug[j]; }
    aug[7] = u; aug[8] = 1.0;
    for (int k = 1; k <= nh; ++k)
        for (int i = 0; i < 7; ++i) {
            double a = 0.0;
            for (int j = 0; j < 9; ++j) a += tab[((k * sb - 1) * 7 + i) * 9 + j] * aug[j];
            S[k][i] = a;
        }
    const double* wt = tab + nb * 63 + 4 * (level - 1);
    const double E = wt[0], wa = wt[1], wm = wt[2], wb = wt[3];
    /* gut */
    const double h = 1.0 / (double)nh;
    double g0 = x[0], g1 = x[1], x2 = x[2], Q = 0.0, R[17], X2[17];
    R[0] = 0.0; X2[0] = x2;
    for (int s = 0; s < nh; ++s) {
        double a0, a1, F1, F2, F3, F4, b0, b1, c0, c1, e0, e1, y0, y1;
        F1 = o_kgut(p, g0 + g1, Dbar) * g1; a0 = -kmax * g0 + d; a1 = kmax * g0 - F1;
        y0 = g0 + 0.5 * h * a0; y1 = g1 + 0.5 * h * a1;
        F2 = o_kgut(p, y0 + y1, Dbar) * y1; b0 = -kmax * y0 + d; b1 = kmax * y0 - F2;
        y0 = g0 + 0.5 * h * b0; y1 = g1 + 0.5 * h * b1;
        F3 = o_kgut(p, y0 + y1, Dbar) * y1; c0 = -kmax * y0 + d; c1 = kmax * y0 - F3;
        y0 = g0 + h * c0; y1 = g1 + h * c1;
        F4 = o_kgut(p, y0 + y1, Dbar) * y1; e0 = -kmax * y0 + d; e1 = kmax * y0 - F4;
        g0 += h / 6.0 * (a0 + 2.0 * b0 + 2.0 * c0 + e0);
        g1 += h / 6.0 * (a1 + 2.0 * b1 + 2.0 * c1 + e1);
        Q += h / 6.0 * (F1 + 2.0 * F2 + 2.0 * F3 + F4);
        x2 = E * x2 + wa * F1 + wm * (0.5 * (F2 + F3)) + wb * F4;
        X2[s + 1] = x2; R[s + 1] = x[2] - x2 + Q;
    }
    /* glucose: x3 itself is carried from step to step (z3 = x3 - c R is formed per step), so that a held x3 stays
     * bit for bit what it was in any precision */
    const double H = 1.0 / (double)ns;
    double x3a = x[3];
    for (int s = 0; s < ns; ++s) {
        const int ia = 2 * s, im = 2 * s + 1, ib = 2 * s + 2;
        const double fz = x3a < 0.0 ? x3a : 0.0;
        y[0] = x3a - c * R[ia];
        o_glucose_rhs(p, y, c * R[ia], c * kabs * X2[ia], S[ia][4], S[ia][6], fz, k1);
        for (int i = 0; i < 3; ++i) w[i] = y[i] + 0.5 * H * k1[i];
        o_glucose_rhs(p, w, c * R[im], c * kabs * X2[im], S[im][4], S[im][6], fz, k2);
        for (int i = 0; i < 3; ++i) w[i] = y[i] + 0.5 * H * k2[i];
        o_glucose_rhs(p, w, c * R[im], c * kabs * X2[im], S[im][4], S[im][6], fz, k3);
        for (int i = 0; i < 3; ++i) w[i] = y[i] + H * k3[i];
        o_glucose_rhs(p, w, c * R[ib], c * kabs * X2[ib], S[ib][4], S[ib][6], fz, k4);
        for (int i = 0; i < 3; ++i) y[i] += H / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        double x3b = y[0] + c * R[ib];
        if (fz < 0.0) x3b = x3a;                                        /* held */
        else if (o_knob[3] > 0.0 && x3b < 0.0) x3b = -o_knob[3];        /* crossed zero in this step */
        x3a = x3b;
    }
    x[0] = g0; x[1] = g1; x[2] = x2;
    x[3] = x3a; x[4] = y[1]; x[12] = y[2];
    for (int j = 0; j < 7; ++j) x[SI[j]] = S[nh][j];
    return level;
}

/* ------------------------------------------------------------------------------------------
 * P2: scipy.integrate.ode(...).set_integrator('dopri5').integrate(t+1)
 *     (patient/t1dpatient.py:276-277 construction, :110-113 one call per minute).
 * Hairer's DOPRI5 driver is re-entered for every minute on [t, t+1]; scipy passes the same
 * work array each time, so only the predicted step size `*h_carry` survives between calls
 * (0 on the first call after reset -> automatic initial-step probe).  Settings as scipy's
 * wrapper passes them: rtol 1e-6, atol 1e-12, nmax 500, safety 0.9, fac1 0.2, fac2 10,
 * beta: the wrapper writes 0.0 into work(5), which the driver reads as "use the default 0.04".
 * hmax = the interval length.  Returns the number of RHS evaluations, or -1 on failure.
 * ---------------------------------------------------------------------------------------- */
static double o_hinit(const double* p, const double* y, const double* f0, double cho, double ins,
                      double lq, double lf, double hmax, double atol, double rtol)
{
    double dnf = 0.0, dny = 0.0, y1[13], f1[13];
    for (int i = 0; i < 13; ++i) {
        const double sk = atol + rtol * fabs(y[i]);
        dnf += (f0[i] / sk) * (f0[i] / sk);
        dny += (y[i] / sk) * (y[i] / sk);
    }
    double h = (dnf <= 1e-10 || dny <= 1e-10) ? 1e-6 : sqrt(dny / dnf) * 0.01;
    if (h > hmax) h = hmax;
    for (int i = 0; i < 13; ++i) y1[i] = y[i] + h * f0[i];
    t1d_o_rhs(p, y1, cho, ins, lq, lf, f1);
    double der2 = 0.0;
    for (int i = 0; i < 13; ++i) {
        const double sk = atol + rtol * fabs(y[i]);
        der2 += ((f1[i] - f0[i]) / sk) * ((f1[i] - f0[i]) / sk);
    }
    der2 = sqrt(der2) / h;
    const double der12 = fmax(fabs(der2), sqrt(dnf));
    double h1 = (der12 <= 1e-15) ? fmax(1e-6, fabs(h) * 1e-3) : pow(0.01 / der12, 1.0 / 5.0);
    h = fmin(fmin(100.0 * fabs(h), h1), hmax);
    return h;
}

int t1d_o_dopri5_minute(const double* p, double* y, double cho, double ins, double lq, double lf,
                        double* h_carry, double beta, double t_start)
{
    static const double a21 = 0.2, a31 = 3.0 / 40.0, a32 = 9.0 / 40.0, a41 = 44.0 / 45.0,
        a42 = -56.0 / 15.0, a43 = 32.0 / 9.0, a51 = 19372.0 / 6561.0, a52 = -25360.0 / 2187.0,
        a53 = 64448.0 / 6561.0, a54 = -212.0 / 729.0, a61 = 9017.0 / 3168.0, a62 = -355.0 / 33.0,
        a63 = 46732.0 / 5247.0, a64 = 49.0 / 176.0, a65 = -5103.0 / 18656.0, a71 = 35.0 / 384.0,
        a73 = 500.0 / 1113.0, a74 = 125.0 / 192.0, a75 = -2187.0 / 6784.0, a76 = 11.0 / 84.0,
        e1 = 71.0 / 57600.0, e3 = -71.0 / 16695.0, e4 = 71.0 / 1920.0, e5 = -17253.0 / 339200.0,
        e6 = 22.0 / 525.0, e7 = -1.0 / 40.0;
    const double rtol = 1e-6, atol = 1e-12, safe = 0.9, fac1 = 0.2, fac2 = 10.0, uround = 2.3e-16;
    const int nmax = 500;
    const double expo1 = 0.2 - beta * 0.75, facc1 = 1.0 / fac1, facc2 = 1.0 / fac2;
    const double xend = t_start + 1.0, hmax = 1.0;
    double x = t_start, h = *h_carry, facold = 1e-4;
    double k1[13], k2[13], k3[13], k4[13], k5[13], k6[13], y1[13], ysti[13];
    int nfcn = 0, nstep = 0, last = 0, reject = 0;

    t1d_o_rhs(p, y, cho, ins, lq, lf, k1); nfcn++;
    if (h == 0.0) { h = o_hinit(p, y, k1, cho, ins, lq, lf, hmax, atol, rtol); nfcn++; }
    for (;;) {
        if (nstep > nmax) return -1;
        if (0.1 * fabs(h) <= fabs(x) * uround) return -1;
        if ((x + 1.01 * h - xend) > 0.0) { h = xend - x; last = 1; }
        nstep++;
        for (int i = 0; i < 13; ++i) y1[i] = y[i] + h * a21 * k1[i];
        t1d_o_rhs(p, y1, cho, ins, lq, lf, k2);
        for (int i = 0; i < 13; ++i) y1[i] = y[i] + h * (a31 * k1[i] + a32 * k2[i]);
        t1d_o_rhs(p, y1, cho, ins, lq, lf, k3);
        for (int i = 0; i < 13; ++i) y1[i] = y[i] + h * (a41 * k1[i] + a42 * k2[i] + a43 * k3[i]);
        t1d_o_rhs(p, y1, cho, ins, lq, lf, k4);
        for (int i = 0; i < 13; ++i) y1[i] = y[i] + h * (a51 * k1[i] + a52 * k2[i] + a53 * k3[i] + a54 * k4[i]);
        t1d_o_rhs(p, y1, cho, ins, lq, lf, k5);
        for (int i = 0; i < 13; ++i) ysti[i] = y[i] + h * (a61 * k1[i] + a62 * k2[i] + a63 * k3[i] + a64 * k4[i] + a65 * k5[i]);
        t1d_o_rhs(p, ysti, cho, ins, lq, lf, k6);
        for (int i = 0; i < 13; ++i) y1[i] = y[i] + h * (a71 * k1[i] + a73 * k3[i] + a74 * k4[i] + a75 * k5[i] + a76 * k6[i]);
        t1d_o_rhs(p, y1, cho, ins, lq, lf, k2);
        nfcn += 6;
        double err = 0.0;
        for (int i = 0; i < 13; ++i) {
            const double ke = (e1 * k1[i] + e3 * k3[i] + e4 * k4[i] + e5 * k5[i] + e6 * k6[i] + e7 * k2[i]) * h;
            const double sk = atol + rtol * fmax(fabs(y[i]), fabs(y1[i]));
            err += (ke / sk) * (ke / sk);
        }
        err = sqrt(err / 13.0);
        const double fac11 = pow(err, expo1);
        double fac = fac11 / pow(facold, beta);
        fac = fmax(facc2, fmin(facc1, fac / safe));
        double hnew = h / fac;
        if (err <= 1.0) {
            facold = fmax(err, 1e-4);
            for (int i = 0; i < 13; ++i) { k1[i] = k2[i]; y[i] = y1[i]; }
            x += h;
            if (last) { *h_carry = hnew; return nfcn; }
            if (fabs(hnew) > hmax) hnew = hmax;
            if (reject) hnew = fmin(fabs(hnew), fabs(h));
            reject = 0;
        } else {
            hnew = h / fmin(facc1, fac11 / safe);
            reject = 1;
            last = 0;
        }
        h = hnew;
    }
}

/* ------------------------------------------------------------------------------------------
 * A1: InsulinPump.basal / .bolus (actuator/pump.py:23-39).  np.round = round-half-to-even,
 * which is rint() under the default rounding mode.  Both use the same arithmetic with their
 * own (inc, min, max).
 * ---------------------------------------------------------------------------------------- */
double t1d_o_pump(double amount, double inc, double lo, double hi)
{
    double v = amount * 6000.0;          /* :24 / :33 */
    v = rint(v / inc) * inc;             /* :25-26 / :34-35 */
    v = v / 6000.0;                      /* :27 / :36 */
    v = (v < hi) ? v : hi;               /* min(v, max)  :28 / :37 */
    v = (v > lo) ? v : lo;               /* max(v, min)  :29 / :38 */
    return v;
}

/* ------------------------------------------------------------------------------------------
 * R1: risk_index([BG], 1) (analysis/risk.py:5-17) for a single value.  Empty-slice means turn
 * into NaN -> nan_to_num -> 0; an infinite value -> DBL_MAX the same way.
 * ---------------------------------------------------------------------------------------- */
void t1d_o_risk(double bg, double* lbgi, double* hbgi, double* ri)
{
    const double f = 1.509 * (pow(log(bg), 1.084) - 5.381);   /* :11 */
    double l = 0.0, hh = 0.0;
    if (f < 0.0) l = 10.0 * f * f;                            /* :12,14 */
    if (f > 0.0) hh = 10.0 * f * f;                           /* :13,15 */
    if (isinf(l)) l = DBL_MAX;
    if (isinf(hh)) hh = DBL_MAX;
    *lbgi = l; *hbgi = hh; *ri = l + hh;                      /* :16 */
}

/* S2: johnson_transform_SU (sensor/noise_gen.py:11-12) */
static double o_johnson(const double* s, double e)
{
    return s[T1D_O_S_XI] + s[T1D_O_S_LAMBDA] * sinh((e - s[T1D_O_S_GAMMA]) / s[T1D_O_S_DELTA]);
}

/* ------------------------------------------------------------------------------------------
 * S2: next(CGMNoise) (sensor/noise_gen.py:61-69 with the refill :30-56 and the AR(1) source
 * :84-97).  State per env: AR value e, the 11 fifteen-minute points of the current block,
 * the count of samples handed out and of normals consumed.  The cubic-spline block is the
 * fixed linear operator W [w_rows x 11] (rows sum to 1; verified against the reference in
 * tests/golden/g4_sensor.npz).  `z` = this env's standard normals, indexed by draw count.
 * ---------------------------------------------------------------------------------------- */
static double o_noise_next(t1d_o_batch* b, int i)
{
    const int n = b->n, S = b->w_rows;
    const int j = b->n_samples[i] % S;
    if (j == 0) {                                   /* deque empty -> _get_noise_seq */
        /* first point = carried-over last point (noise_gen.py:33-36) */
        if (b->n_samples[i] > 0) b->pts[0 * n + i] = b->pts[10 * n + i];
        for (int k = 1; k <= 10; ++k) {             /* :34  ten new 15-min points */
            const double z = b->normals[(size_t)b->n_draws[i] * n + i];
            b->n_draws[i]++;
            b->ar_e[i] = b->sensor[T1D_O_S_PACF] * (b->ar_e[i] + z);   /* :88 */
            b->pts[k * n + i] = o_johnson(b->sensor, b->ar_e[i]);
        }
    }
    double acc = 0.0;
    for (int k = 0; k < 11; ++k) acc += b->W[j * 11 + k] * b->pts[k * n + i];
    b->n_samples[i]++;
    return acc;
}

/* S1: CGMSensor.measure (sensor/cgm.py:26-36) at patient time t (minutes, after the step) */
static double o_measure(t1d_o_batch* b, int i, double gsub)
{
    const double st = b->sensor[T1D_O_S_SAMPLE_TIME];
    if (fmod((double)b->t[i], st) == 0.0) {
        double cgm = gsub + o_noise_next(b, i);
        cgm = fmax(cgm, b->sensor[T1D_O_S_MIN]);
        cgm = fmin(cgm, b->sensor[T1D_O_S_MAX]);
        b->last_cgm[i] = cgm;
        return cgm;
    }
    return b->last_cgm[i];
}

/* ------------------------------------------------------------------------------------------
 * E3/P4/S1: T1DSimEnv.reset() (simulation/env.py:119-155) on every env of the batch:
 * patient.reset (patient/t1dpatient.py:247-281, x0 = columns x0_1..x0_13, or the caller's
 * init_state when x0_override != NULL -- how random_init_bg draws made on the host enter),
 * sensor.reset (sensor/cgm.py:47-50: new noise generator => consumes normal #0),
 * _reset (env.py:119-134: BG0, risk, CGM sample #0 -> history[0]) and the returned
 * observation = CGM sample #1 (env.py:142).
 * ---------------------------------------------------------------------------------------- */
void t1d_o_reset(t1d_o_batch* b, const double* x0_override, t1d_o_out* o)
{
    const int n = b->n;
    for (int i = 0; i < n; ++i) {
        const double* p = b->ptab + (size_t)b->pid[i] * T1D_O_NPAR;
        for (int k = 0; k < 13; ++k)
            b->x[k * n + i] = x0_override ? x0_override[k * n + i] : p[T1D_O_X0 + k];
        b->planned[i] = 0.0;
        b->last_qsto[i] = b->x[0 * n + i] + b->x[1 * n + i];    /* t1dpatient.py:272 */
        b->last_food[i] = 0.0;
        b->was_eating[i] = 0;
        b->t[i] = 0;
        b->h_carry[i] = 0.0;
        /* sensor */
        b->last_cgm[i] = 0.0; b->n_samples[i] = 0; b->n_draws[i] = 0;
        b->ar_e[i] = b->normals[(size_t)0 * n + i]; b->n_draws[i] = 1;          /* noise_gen.py:86 */
        for (int k = 0; k < 11; ++k) b->pts[k * n + i] = 0.0;
        b->pts[0 * n + i] = o_johnson(b->sensor, b->ar_e[i]);                  /* noise_gen.py:24 */
        const double bg0 = b->x[12 * n + i] / p[T1D_O_VG];
        const double cgm0 = o_measure(b, i, bg0);        /* env.py:126 */
        const double cgm1 = o_measure(b, i, bg0);        /* env.py:142 */
        b->prev_cgm[i] = cgm0;                           /* CGM_hist[0] */
        if (o) {
            o->cgm[i] = cgm1; o->bg[i] = bg0; o->reward[i] = 0.0; o->done[i] = 0; o->meal[i] = 0.0;
            o->insulin[i] = 0.0; o->cgm_hist0[i] = cgm0;
            t1d_o_risk(bg0, &o->lbgi[i], &o->hbgi[i], &o->risk[i]);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * E2: T1DSimEnv.step (simulation/env.py:66-117) = int(sample_time) x E1 mini_step
 * (env.py:48-64), with P3 T1DPatient.step bookkeeping (patient/t1dpatient.py:82-107,222-236),
 * the default reward risk_diff (env.py:27-33) on the CGM history, and done (env.py:103).
 * cho[m*n+i] = grams the scenario announces to env i in minute m of this step.
 * integrator: 0 = RK4(n_sub), 1 = DOPRI5 as SciPy runs it (beta in `dopri_beta`).
 * Returns 0, or -1 if the adaptive solver failed for some env.
 * ---------------------------------------------------------------------------------------- */
int t1d_o_step(t1d_o_batch* b, const double* basal, const double* bolus, const double* cho,
               int integrator, int n_sub, double dopri_beta, t1d_o_out* o)
{
    const int n = b->n;
    const double st = b->sensor[T1D_O_S_SAMPLE_TIME];
    const int nmin = (int)st;
    int rc = 0;
    for (int i = 0; i < n; ++i) {
        const double* p = b->ptab + (size_t)b->pid[i] * T1D_O_NPAR;
        double x[13];
        for (int k = 0; k < 13; ++k) x[k] = b->x[k * n + i];
        double a_cho = 0.0, a_ins = 0.0, a_bg = 0.0, a_cgm = 0.0;
        const double q_basal = t1d_o_pump(basal[i], b->pump[T1D_O_PU_INC_BASAL], b->pump[T1D_O_PU_MIN_BASAL], b->pump[T1D_O_PU_MAX_BASAL]);
        const double q_bolus = t1d_o_pump(bolus ? bolus[i] : 0.0, b->pump[T1D_O_PU_INC_BOLUS], b->pump[T1D_O_PU_MIN_BOLUS], b->pump[T1D_O_PU_MAX_BOLUS]);
        const double insulin = q_basal + q_bolus;                       /* env.py:51-53 */
        for (int m = 0; m < nmin; ++m) {
            const double meal = cho ? cho[(size_t)m * n + i] : 0.0;     /* env.py:50,54 */
            /* _announce_meal (t1dpatient.py:222-236) */
            double to_eat = 0.0;
            b->planned[i] += meal;
            if (b->planned[i] > 0.0) {
                to_eat = fmin(5.0, b->planned[i]);
                b->planned[i] -= to_eat;
                b->planned[i] = fmax(0.0, b->planned[i]);
            }
            /* eating edges (t1dpatient.py:88-107) */
            if (to_eat > 0.0 && !b->was_eating[i]) { b->last_qsto[i] = x[0] + x[1]; b->last_food[i] = 0.0; }
            b->last_food[i] += to_eat;          /* is_eating <=> to_eat > 0 after the edge test */
            b->was_eating[i] = (to_eat > 0.0);
            if (integrator == 0) {
                t1d_o_rk4_minute(p, x, to_eat, insulin, b->last_qsto[i], b->last_food[i], n_sub);
            } else if (integrator == 2) {
                t1d_o_mr_minute(p, x, to_eat, insulin, b->last_qsto[i], b->last_food[i], n_sub / 1000, n_sub % 1000);
            } else if (integrator == 3) {
                if (t1d_o_split_minute(p, b->split_tab ? b->split_tab + (size_t)b->pid[i] * b->split_stride : NULL, x,
                                       to_eat, insulin, b->last_qsto[i], b->last_food[i], n_sub, 0) < 0) rc = -1;
                if (b->level_count) b->level_count[1]++;
            } else if (integrator == 4) {
                const int lv = t1d_o_split_minute(p, b->split_tab ? b->split_tab + (size_t)b->pid[i] * b->split_stride : NULL, x,
                                                  to_eat, insulin, b->last_qsto[i], b->last_food[i], n_sub, 1);
                if (lv < 0) rc = -1; else if (b->level_count) b->level_count[lv]++;
            } else {
                if (t1d_o_dopri5_minute(p, x, to_eat, insulin, b->last_qsto[i], b->last_food[i],
                                        &b->h_carry[i], dopri_beta, (double)b->t[i]) < 0) rc = -1;
            }
            b->t[i] += 1;
            const double bg = x[12] / p[T1D_O_VG];                      /* env.py:61 */
            const double cgm = o_measure(b, i, bg);                     /* env.py:62 */
            a_cho += meal / st; a_ins += insulin / st; a_bg += bg / st; a_cgm += cgm / st;  /* env.py:78-81 */
        }
        for (int k = 0; k < 13; ++k) b->x[k * n + i] = x[k];
        double rp, rcur, l, hh;
        t1d_o_risk(b->prev_cgm[i], &l, &hh, &rp);                       /* env.py:27-33 */
        t1d_o_risk(a_cgm, &l, &hh, &rcur);
        if (o) {
            o->cgm[i] = a_cgm; o->bg[i] = a_bg; o->meal[i] = a_cho; o->insulin[i] = a_ins;
            o->reward[i] = rp - rcur;
            o->done[i] = (a_bg < 70.0 || a_bg > 350.0);                 /* env.py:103 */
            t1d_o_risk(a_bg, &o->lbgi[i], &o->hbgi[i], &o->risk[i]);    /* env.py:85 */
        }
        b->prev_cgm[i] = a_cgm;
    }
    return rc;
}

/* C1: PIDController.policy (controller/pid_ctrller.py:17-36) -> basal U/min */
double t1d_o_pid(double* integ, double* prev, double cgm, double P, double I, double D, double target,
                 double sample_time)
{
    const double u = P * (cgm - target) + I * (*integ) + D * (cgm - *prev) / sample_time;  /* :22-24 */
    *prev = cgm;                                    /* :29 */
    *integ += (cgm - target) * sample_time;          /* :30 */
    return u;
}

/* Patient-only minute: T1DPatient.step(Action(CHO, insulin)) without env/sensor/pump, used by
 * the G2 open-loop pins (patient/t1dpatient.py:82-116). */
int t1d_o_patient_minute(const double* p, double* x, double* planned, double* last_qsto, double* last_food,
                         uint8_t* was_eating, double* h_carry, int t, double meal, double insulin,
                         int integrator, int n_sub, double dopri_beta, const double* split_tab)
{
    double to_eat = 0.0;
    *planned += meal;
    if (*planned > 0.0) { to_eat = fmin(5.0, *planned); *planned -= to_eat; *planned = fmax(0.0, *planned); }
    if (to_eat > 0.0 && !*was_eating) { *last_qsto = x[0] + x[1]; *last_food = 0.0; }
    *last_food += to_eat;
    *was_eating = (to_eat > 0.0);
    if (integrator == 0) { t1d_o_rk4_minute(p, x, to_eat, insulin, *last_qsto, *last_food, n_sub); return 4 * n_sub; }
    if (integrator == 2) { t1d_o_mr_minute(p, x, to_eat, insulin, *last_qsto, *last_food, n_sub / 1000, n_sub % 1000); return 4 * (n_sub / 1000); }
    if (integrator == 3 || integrator == 4)
        return t1d_o_split_minute(p, split_tab, x, to_eat, insulin, *last_qsto, *last_food, n_sub, integrator == 4) < 0 ? -1 : 4 * n_sub;
    return t1d_o_dopri5_minute(p, x, to_eat, insulin, *last_qsto, *last_food, h_carry, dopri_beta, (double)t);
}
