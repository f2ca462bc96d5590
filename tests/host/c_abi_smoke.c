/* Plain C99 caller of libemei_hip.so: proves that include/emei_hip.h is a C header (no C++ / torch types in
 * any signature) and that the library is usable without Python.  Built by tests/test_c_abi_program.py with
 *   gcc -std=c99 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ ... -lemei_hip -lamdhip64
 * Run on a GPU box it rolls 4096 CartPoleSwingUp envs for 32 steps from the device reset, twice, and checks
 * that both runs agree and that the outputs are sane. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "emei_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 2; } } while (0)
#define CHECK_EMEI(x) do { int r_ = (x); if (r_ != EMEI_OK) { printf("emei error %d (%s) at %s\n", r_, emei_last_error(), #x); return 3; } } while (0)

int main(int argc, char** argv) {
    if (emei_abi_version() != EMEI_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }
    int od = 0, ad = 0, sd = 0;
    CHECK_EMEI(emei_env_dims(EMEI_CARTPOLE_SWINGUP, &od, &ad, &sd));
    if (od != 4 || ad != 0 || sd != 4) { printf("bad dims\n"); return 1; }
    if (argc > 1 && strcmp(argv[1], "--link-only") == 0) { printf("LINK OK\n"); return 0; }

    const int64_t n = 4096;
    const int T = 32;
    emei_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg), cfg.env_id = EMEI_CARTPOLE_SWINGUP, cfg.n_envs = n, cfg.freq_rate = 1;
    cfg.precision = EMEI_PRECISION_REF, cfg.real_time_scale = 0.02, cfg.max_episode_steps = 1000, cfg.device = 0, cfg.seed = 7;
    unsigned char *act_h = (unsigned char*)malloc((size_t)T * n), *act_d = NULL, *done_d = NULL;
    float *obs_d = NULL, *rew_d = NULL, *obs_h[2];
    for (size_t i = 0; i < (size_t)T * n; ++i) act_h[i] = (unsigned char)((i * 2654435761u >> 13) & 1u);
    CHECK_HIP(hipMalloc((void**)&act_d, (size_t)T * n));
    CHECK_HIP(hipMalloc((void**)&obs_d, (size_t)T * n * 4 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&rew_d, (size_t)T * n * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&done_d, (size_t)T * n));
    CHECK_HIP(hipMemcpy(act_d, act_h, (size_t)T * n, hipMemcpyHostToDevice));
    for (int run = 0; run < 2; ++run) {
        emei_env* h = NULL;
        CHECK_EMEI(emei_create(&cfg, &h));
        /* step before reset must be refused with the reference's assertion text (base_control.py:67) */
        if (emei_step(h, act_d, EMEI_ACT_U8, obs_d, rew_d, done_d, 0, NULL) != EMEI_ERR_STATE) { printf("no state check\n"); return 1; }
        CHECK_EMEI(emei_reset(h, 7, NULL));
        CHECK_EMEI(emei_rollout(h, T, act_d, EMEI_ACT_U8, obs_d, rew_d, done_d, EMEI_FLAG_AUTO_RESET, NULL));
        CHECK_HIP(hipDeviceSynchronize());
        obs_h[run] = (float*)malloc((size_t)T * n * 4 * sizeof(float));
        CHECK_HIP(hipMemcpy(obs_h[run], obs_d, (size_t)T * n * 4 * sizeof(float), hipMemcpyDeviceToHost));
        CHECK_EMEI(emei_destroy(h));
    }
    if (memcmp(obs_h[0], obs_h[1], (size_t)T * n * 4 * sizeof(float)) != 0) { printf("runs differ\n"); return 1; }
    double max_x = 0, mean_th = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float* o = obs_h[0] + ((size_t)(T - 1) * n + i) * 4;
        if (!isfinite(o[0]) || !isfinite(o[1]) || !isfinite(o[2]) || !isfinite(o[3])) { printf("non-finite obs\n"); return 1; }
        if (fabs(o[0]) > max_x) max_x = fabs(o[0]);
        mean_th += o[2] / n;
    }
    if (!(max_x < 5.0) || !(fabs(mean_th - 3.14159) < 0.5)) { printf("implausible: max|x| %.3f mean theta %.3f\n", max_x, mean_th); return 1; }
    /* the gym single-env call with host values (emei_step_host): 64 steps of one env through page-locked host memory must equal
     * the same env stepped with emei_step on device buffers, bit for bit, and the float64 observation must be the state's */
    {
        cfg.n_envs = 1;
        emei_env *ha = NULL, *hb = NULL;
        CHECK_EMEI(emei_create(&cfg, &ha));
        CHECK_EMEI(emei_create(&cfg, &hb));
        CHECK_EMEI(emei_reset(ha, 11, NULL));
        CHECK_EMEI(emei_reset(hb, 11, NULL));
        unsigned char* pin = NULL;
        CHECK_HIP(hipHostMalloc((void**)&pin, 256, 0));
        unsigned char* a_p = pin;            /* action */
        double* o64_p = (double*)(pin + 64); /* 4 doubles */
        float* o32_p = (float*)(pin + 128);
        float* r_p = (float*)(pin + 160);
        unsigned char* d_p = pin + 192;
        double state[4];
        double* state_d = NULL;
        CHECK_HIP(hipMalloc((void**)&state_d, sizeof(state)));
        if (emei_step_host(ha, a_p, EMEI_ACT_U8, NULL, o32_p, r_p, d_p, 0, NULL) != EMEI_ERR_INVALID) { printf("null check\n"); return 1; }
        for (int t = 0; t < 64; ++t) {
            *a_p = act_h[t];
            CHECK_EMEI(emei_step_host(ha, a_p, EMEI_ACT_U8, o64_p, o32_p, r_p, d_p, 0, NULL));
            float o_ref[4], r_ref;
            unsigned char d_ref;
            CHECK_HIP(hipMemcpy(act_d, a_p, 1, hipMemcpyHostToDevice));
            CHECK_EMEI(emei_step(hb, act_d, EMEI_ACT_U8, obs_d, rew_d, done_d, 0, NULL));
            CHECK_EMEI(emei_get_state(hb, state_d, NULL));
            CHECK_HIP(hipDeviceSynchronize());
            CHECK_HIP(hipMemcpy(o_ref, obs_d, sizeof(o_ref), hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(&r_ref, rew_d, sizeof(r_ref), hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(&d_ref, done_d, 1, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(state, state_d, sizeof(state), hipMemcpyDeviceToHost));
            if (memcmp(o_ref, o32_p, sizeof(o_ref)) || memcmp(&r_ref, r_p, 4) || d_ref != *d_p || memcmp(state, o64_p, sizeof(state))) {
                printf("emei_step_host differs from emei_step at step %d\n", t);
                return 1;
            }
        }
        CHECK_EMEI(emei_destroy(ha));
        CHECK_EMEI(emei_destroy(hb));
        CHECK_HIP(hipHostFree(pin));
    }
    printf("C ABI OK: %lld envs x %d steps, max|x| %.3f, mean theta %.3f\n", (long long)n, T, max_x, mean_th);
    return 0;
}
