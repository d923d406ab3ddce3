# bench.py's preset legs alone (CFEAR_BENCH_PRESETS=name,name; CFEAR_PRESET_B=sequences of every selected leg, 0 = the bench's own): for A/B
# runs and PMC passes of one preset without the ten minutes of the whole bench line
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    torch.cuda.set_stream(torch.cuda.Stream())
    from cfear_radarodometry_code_public_amd import capi
    dev = torch.device("cuda:0")
    W, K = 4, 12
    streams = bench.make_streams(16, bench.PRE_ROLL + W + K, 0)
    Bs = [int(v) for v in os.environ.get("CFEAR_PRESET_B", "0").split(",")]
    for B in Bs:
        if B:
            os.environ["CFEAR_PRESET_FORCE_B"] = str(B)
        out = bench.preset_legs(torch, dev, capi, 0, torch.cuda.current_stream().cuda_stream, streams, None, W=W, K=K)
        for name, r in out.items():
            print("%s B=%d: %.0f scans/s  filter %.0f  features %.0f  registration %.0f us  cells %d residuals %d kf %d inner %s" %
                  (name, r["sequences"], r["scans_per_s"], r["kstrongest_launch_us"], r["features_launch_us"], r["registration_launch_us"], r["cells_seq0"],
                   r["residuals_seq0"], r["keyframes_seq0"], r["inner_iterations_seq0"]), flush=True)


if __name__ == "__main__":
    main()
