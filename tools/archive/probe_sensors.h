/* GPU sensors of the render node(s) this container may open (hwmon: sclk, mclk, socket power, junction / HBM temperature).
 * Shared by tools/probe_proc.c and tools/probe_sustain.c. */
#ifndef MA_PROBE_SENSORS_H
#define MA_PROBE_SENSORS_H
#include <dirent.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static long read_long(const char* path) {
    char buf[64];
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    ssize_t k = read(fd, buf, sizeof buf - 1);
    close(fd);
    if (k <= 0) return -1;
    buf[k] = 0;
    return strtol(buf, NULL, 10);
}

/* sensors of the GPU this container was given: the render nodes it may open */
static char g_hwmon[8][300];
static int g_n_hwmon = 0;

static void find_hwmon(void) {
    DIR* d = opendir("/dev/dri");
    struct dirent* e;
    while (d && (e = readdir(d)) && g_n_hwmon < 8) {
        if (strncmp(e->d_name, "renderD", 7) != 0) continue;
        char dev[300], hw[300];
        snprintf(dev, sizeof dev, "/dev/dri/%s", e->d_name);
        if (access(dev, R_OK | W_OK) != 0) continue;
        snprintf(hw, sizeof hw, "/sys/class/drm/%s/device/hwmon", e->d_name);
        DIR* h = opendir(hw);
        struct dirent* he;
        while (h && (he = readdir(h))) {
            if (strncmp(he->d_name, "hwmon", 5) != 0) continue;
            snprintf(g_hwmon[g_n_hwmon++], sizeof g_hwmon[0], "%.200s/%.60s", hw, he->d_name);
            break;
        }
        if (h) closedir(h);
    }
    if (d) closedir(d);
}

static void print_sensors(void) {
    for (int i = 0; i < g_n_hwmon; ++i) {
        char p[400];
        const char* names[] = {"freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input"};
        const char* keys[] = {"sclk_hz", "mclk_hz", "power_avg_uw", "power_uw", "temp_edge_mc", "temp_junction_mc", "temp_mem_mc"};
        for (int k = 0; k < 7; ++k) {
            snprintf(p, sizeof p, "%s/%s", g_hwmon[i], names[k]);
            long v = read_long(p);
            if (v >= 0) printf(", \"%s%s\": %ld", keys[k], i ? "_b" : "", v);
        }
    }
}

#endif
