/* power_sampler.c -- socket power / shader clock trace of GPU 0 while another process runs a kernel.
 *
 *   tools/power_sampler <seconds> <hz> > trace.csv        (tools/gpu_power_trace.sh drives it)
 *
 * Reads the SMU's gpu_metrics table through librocm_smi64 (the same source rocm-smi prints from) at <hz> samples
 * per second: current socket power, the per-XCD shader clocks, the energy accumulator, GFX activity and the firmware's
 * RESIDENCY counters -- ppt_residency_acc / accumulation_counter is the share of time the power limiter (PPT) was
 * the active clock constraint, which is the direct form of DESIGN.md 5.1.3's "the matrix pipes are power-limited".
 * Diagnostic tool: not part of the product path.
 *
 * gcc -O2 -I/opt/rocm/include tools/power_sampler.c -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib -o tools/power_sampler
 */
#include <rocm_smi/rocm_smi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const double hz = argc > 2 ? atof(argv[2]) : 20.0;
    if (rsmi_init(0) != RSMI_STATUS_SUCCESS) {
        fprintf(stderr, "rsmi_init failed\n");
        return 1;
    }
    uint64_t cap = 0;
    if (rsmi_dev_power_cap_get(0, 0, &cap) == RSMI_STATUS_SUCCESS) printf("# power cap %.0f W\n", cap * 1e-6);
    rsmi_frequencies_t f;
    if (rsmi_dev_gpu_clk_freq_get(0, RSMI_CLK_TYPE_SYS, &f) == RSMI_STATUS_SUCCESS && f.num_supported > 0)
        printf("# sclk levels: min %.0f MHz, max %.0f MHz\n", f.frequency[0] * 1e-6, f.frequency[f.num_supported - 1] * 1e-6);
    printf("t_s,socket_power_W,gfxclk_min_MHz,gfxclk_max_MHz,gfxclk_mean_MHz,avg_gfxclk_MHz,gfx_activity_pct,temp_hotspot_C,"
           "energy_acc,accumulation_counter,ppt_residency_acc,prochot_residency_acc,socket_thm_residency_acc,"
           "vr_thm_residency_acc,hbm_thm_residency_acc,throttle_status,indep_throttle_status,uclk_MHz\n");
    const double t0 = now_s();
    for (long k = 0;; ++k) {
        const double t = now_s() - t0;
        if (t > seconds) break;
        rsmi_gpu_metrics_t m;
        memset(&m, 0, sizeof(m));
        if (rsmi_dev_gpu_metrics_info_get(0, &m) == RSMI_STATUS_SUCCESS) {
            unsigned lo = 65535, hi = 0, n = 0;
            double sum = 0;
            for (int x = 0; x < RSMI_MAX_NUM_GFX_CLKS; ++x) {
                const unsigned c = m.current_gfxclks[x];
                if (c == 0 || c == 65535) continue;
                lo = c < lo ? c : lo;
                hi = c > hi ? c : hi;
                sum += c;
                ++n;
            }
            uint64_t pw = 0;
            double watts = m.current_socket_power != 65535 ? (double)m.current_socket_power : -1.0;
            if (watts < 0 && rsmi_dev_current_socket_power_get(0, &pw) == RSMI_STATUS_SUCCESS) watts = pw * 1e-6;
            printf("%.4f,%.1f,%u,%u,%.1f,%u,%u,%u,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%u,%llu,%u\n", t, watts, n ? lo : 0, hi,
                   n ? sum / n : 0.0, (unsigned)m.average_gfxclk_frequency, (unsigned)m.average_gfx_activity,
                   (unsigned)m.temperature_hotspot, (unsigned long long)m.energy_accumulator,
                   (unsigned long long)m.accumulation_counter, (unsigned long long)m.ppt_residency_acc,
                   (unsigned long long)m.prochot_residency_acc, (unsigned long long)m.socket_thm_residency_acc,
                   (unsigned long long)m.vr_thm_residency_acc, (unsigned long long)m.hbm_thm_residency_acc,
                   (unsigned)m.throttle_status, (unsigned long long)m.indep_throttle_status, (unsigned)m.current_uclk);
        }
        fflush(stdout);
        const double next = t0 + (k + 1) / hz;
        const double d = next - now_s();
        if (d > 0) {
            struct timespec ts = {(time_t)d, (long)((d - (long)d) * 1e9)};
            nanosleep(&ts, NULL);
        }
    }
    rsmi_shut_down();
    return 0;
}
