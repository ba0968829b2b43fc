/* t1d_oracle.h -- interface of the CPU oracle (TEST INFRASTRUCTURE ONLY; see t1d_oracle.c). */
#ifndef T1D_ORACLE_H
#define T1D_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* columns of one patient row (the host builds it from params/vpatient_params.csv) */
enum {
    T1D_O_X0 = 0,      /* 13 entries: x0_1 .. x0_13 */
    T1D_O_BW = 13, T1D_O_KABS, T1D_O_KMAX, T1D_O_KMIN, T1D_O_B, T1D_O_D, T1D_O_VG, T1D_O_VI,
    T1D_O_VMX, T1D_O_KM0, T1D_O_K2, T1D_O_K1, T1D_O_P2U, T1D_O_M1, T1D_O_M2, T1D_O_M4, T1D_O_M30,
    T1D_O_IB, T1D_O_KI, T1D_O_KP2, T1D_O_KP3, T1D_O_F, T1D_O_KE1, T1D_O_KE2, T1D_O_FSNC,
    T1D_O_VM0, T1D_O_KD, T1D_O_KSC, T1D_O_KA1, T1D_O_KA2, T1D_O_KP1, T1D_O_U2SS,
    T1D_O_NPAR                                         /* = 45 */
};
/* sensor row (params/sensor_params.csv) */
enum { T1D_O_S_PACF = 0, T1D_O_S_GAMMA, T1D_O_S_LAMBDA, T1D_O_S_DELTA, T1D_O_S_XI,
       T1D_O_S_SAMPLE_TIME, T1D_O_S_MIN, T1D_O_S_MAX, T1D_O_S_N };
/* pump row (params/pump_params.csv) */
enum { T1D_O_PU_MIN_BOLUS = 0, T1D_O_PU_MAX_BOLUS, T1D_O_PU_INC_BOLUS, T1D_O_PU_MIN_BASAL,
       T1D_O_PU_MAX_BASAL, T1D_O_PU_INC_BASAL, T1D_O_PU_N };

typedef struct {
    int32_t n;                 /* envs */
    int32_t w_rows;            /* samples per 150-min noise block */
    const double* ptab;        /* [n_patients][T1D_O_NPAR] */
    const int32_t* pid;        /* [n] row of ptab */
    const double* W;           /* [w_rows][11] spline block operator */
    const double* normals;     /* [n_draws_max][n] standard normals per env, in draw order */
    double sensor[T1D_O_S_N];
    double pump[T1D_O_PU_N];
    /* state, struct-of-arrays, env index fastest */
    double* x;                 /* [13][n] */
    double* planned;           /* [n] planned_meal (g) */
    double* last_qsto;         /* [n] mg */
    double* last_food;         /* [n] g */
    uint8_t* was_eating;       /* [n] last eaten CHO > 0 */
    int32_t* t;                /* [n] minutes since episode start */
    double* h_carry;           /* [n] DOPRI5 predicted step carried between minutes */
    double* last_cgm;          /* [n] zero-order hold */
    double* ar_e;              /* [n] AR(1) state */
    double* pts;               /* [11][n] Johnson-SU points of the current block */
    int32_t* n_samples;        /* [n] noise samples handed out */
    int32_t* n_draws;          /* [n] normals consumed */
    double* prev_cgm;          /* [n] CGM_hist[-1] before this step (default reward) */
    const double* split_tab;   /* [n_patients][split_stride] tables of the split scheme (integrators 3, 4) or NULL */
    int32_t split_stride;      /* = 126 n_sub + 8 */
    int64_t* level_count;      /* [3] or NULL: env-minutes integrated at level 1 / 2 of the split scheme ([0] unused; studies) */
} t1d_o_batch;

typedef struct {
    double *cgm, *bg, *reward, *lbgi, *hbgi, *risk, *meal, *insulin, *cgm_hist0;   /* [n] each */
    uint8_t* done;                                                                  /* [n] */
} t1d_o_out;

void t1d_o_rhs(const double* p, const double* x, double cho, double ins, double lq, double lf, double* dx);
void t1d_o_rk4_minute(const double* p, double* x, double cho, double ins, double lq, double lf, int n_sub);
void t1d_o_mr_minute(const double* p, double* x, double cho, double ins, double lq, double lf, int ng, int ns);
int t1d_o_split_minute(const double* p, const double* tab, double* x, double cho, double ins, double lq,
                       double lf, int n_sub, int mode);
void t1d_o_set_knob(int k, double v);
double t1d_o_get_knob(int k);
int t1d_o_dopri5_minute(const double* p, double* y, double cho, double ins, double lq, double lf,
                        double* h_carry, double beta, double t_start);
double t1d_o_pump(double amount, double inc, double lo, double hi);
void t1d_o_risk(double bg, double* lbgi, double* hbgi, double* ri);
void t1d_o_reset(t1d_o_batch* b, const double* x0_override, t1d_o_out* o);
int t1d_o_step(t1d_o_batch* b, const double* basal, const double* bolus, const double* cho,
               int integrator, int n_sub, double dopri_beta, t1d_o_out* o);
double t1d_o_pid(double* integ, double* prev, double cgm, double P, double I, double D, double target,
                 double sample_time);
int t1d_o_patient_minute(const double* p, double* x, double* planned, double* last_qsto, double* last_food,
                         uint8_t* was_eating, double* h_carry, int t, double meal, double insulin,
                         int integrator, int n_sub, double dopri_beta, const double* split_tab);

#ifdef __cplusplus
}
#endif
#endif
