#include <cstdio>
#include "hydrochrono_amd_yaml.h"
int main(int argc, char** argv) {
    const char* sfields[] = {"wave_type", "type", "spectrum", "x"};
    for (int i = 1; i < argc; ++i) {
        hc_yaml* cfg = nullptr; char err[512];
        int rc = hc_yaml_read(argv[i], &cfg, err, sizeof err);
        if (rc) { std::printf("%s: error %d\n", argv[i], rc); continue; }
        int nb = hc_yaml_num_bodies(cfg);
        for (int b = 0; b < nb + 1; ++b) { hc_yaml_body_string(cfg, b, "name"); hc_yaml_body_string(cfg, b, "h5_file"); hc_yaml_body_number(cfg, b, "nope"); }
        for (auto f : sfields) { hc_yaml_string(cfg, f); hc_yaml_number(cfg, f); }
        double buf[4]; hc_yaml_period_values(cfg, buf, 4); hc_yaml_period_values(cfg, nullptr, 0);
        std::printf("%s: ok bodies=%d\n", argv[i], nb);
        hc_yaml_free(cfg);
    }
    hc_yaml* cfg = nullptr; char e2[8];
    hc_yaml_read("/nonexistent.yaml", &cfg, e2, sizeof e2);
    return 0;
}
