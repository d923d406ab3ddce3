// TEMPORARY during bring-up: entry points not implemented yet fail loudly.
#include "common.h"
#define NI(ctx) cfear_fail((ctx), CFEAR_ERR_UNSUPPORTED, "not implemented yet")
extern "C" {
int cfear_filter_polar(cfear_ctx* c, const uint8_t*, cfear_cloud**, cfear_cloud**) { return NI(c); }
int cfear_filter_polar_device(cfear_ctx* c, const uint8_t*, cfear_cloud**, cfear_cloud**) { return NI(c); }
int cfear_cloud_upload(cfear_ctx* c, const float*, int, cfear_cloud**) { return NI(c); }
int cfear_cloud_size(cfear_ctx* c, const cfear_cloud*, int*) { return NI(c); }
int cfear_cloud_download(cfear_ctx* c, const cfear_cloud*, float*, int, int*) { return NI(c); }
void cfear_cloud_release(cfear_ctx*, cfear_cloud*) {}
int cfear_compensate(cfear_ctx* c, cfear_cloud*, const double*, int) { return NI(c); }
int cfear_scan_create(cfear_ctx* c, const cfear_cloud*, cfear_scan**) { return NI(c); }
void cfear_scan_release(cfear_ctx*, cfear_scan*) {}
int cfear_scan_size(cfear_ctx* c, const cfear_scan*, int*) { return NI(c); }
int cfear_scan_download_cells(cfear_ctx* c, const cfear_scan*, cfear_cell*, int, int*) { return NI(c); }
int cfear_scan_closest(cfear_ctx* c, const cfear_scan*, const double*, int, double, int32_t*) { return NI(c); }
int cfear_register(cfear_ctx* c, cfear_scan* const*, int, double*, double*, cfear_reg_summary*) { return NI(c); }
int cfear_odometry_create(cfear_ctx* c, int, cfear_odometry**) { return NI(c); }
void cfear_odometry_destroy(cfear_ctx*, cfear_odometry*) {}
int cfear_odometry_reset(cfear_ctx* c, cfear_odometry*) { return NI(c); }
int cfear_odometry_step_device(cfear_ctx* c, cfear_odometry*, const uint8_t*) { return NI(c); }
int cfear_odometry_step_host(cfear_ctx* c, cfear_odometry*, const uint8_t*) { return NI(c); }
int cfear_odometry_poses(cfear_ctx* c, cfear_odometry*, double*) { return NI(c); }
int cfear_odometry_summary(cfear_ctx* c, cfear_odometry*, int, cfear_reg_summary*, int*, int*) { return NI(c); }
}
