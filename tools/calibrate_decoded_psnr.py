"""What the REFERENCE DTYPE costs in decoded pixels over 50 guided steps: the final latents of the all-bf16 oracle run (tools/make_fullsize_golden_gpu.py calib_full ->
gpurun_out/fixtures/bf16_full_oracle_<task>50_final_latents.npz) decoded by the SAME fp32 CPU oracle VAE that decoded the fp32 fixtures, rgb PSNR / disparity rel-L2 against the
fixtures' decoded clips (every 8th row / column).  The native path's 36.5 / 36.3 dB (tests/test_fullsize_guided_gpu.py) are to be read against these numbers, not against the decoder's own
49.5 dB on identical latents.  ~20 min of CPU per task; writes profiles/r05_bf16_oracle_decoded_psnr.json.

    python tools/calibrate_decoded_psnr.py prediction planning
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fullsize_cases as fc  # noqa: E402


def main():
    tasks = sys.argv[1:] or ["prediction", "planning"]
    torch.set_num_threads(int(os.environ.get("AETHER_GOLDEN_THREADS", os.cpu_count() or 8)))
    vae = fc.build_oracle_vae()
    sf = vae.config.scaling_factor
    out_path = os.path.join(fc.ROOT, "profiles", "r05_bf16_oracle_decoded_psnr.json")
    res = json.load(open(out_path)) if os.path.exists(out_path) else {"case": "final latents of the all-bf16 ORACLE after 50 guided steps, decoded by the fp32 CPU oracle VAE, against the fp32 "
                                                                             "fixtures' decoded clips (every 8th row / column)"}
    s = fc.DEC_STRIDE
    for task in tasks:
        if task.startswith("recon"):                                   # recon4 / recon50: the reconstruction trajectories under device semantics (calib_recon)
            z = np.load(os.path.join(fc.GOLDEN_DIR, f"fullsize_{task}_device.npz"))
            lat = fc.from_bf16_bits(np.load(os.path.join(fc.ROOT, "gpurun_out", "fixtures", f"bf16_oracle_{task}_final_latents.npz"))["final_latents_bits"])
        else:
            z = np.load(os.path.join(fc.GOLDEN_DIR, f"fullsize_{task}50.npz"))
            lat = fc.from_bf16_bits(np.load(os.path.join(fc.ROOT, "gpurun_out", "fixtures", f"bf16_full_oracle_{task}50_final_latents.npz"))["final_latents_bits"])
        t0 = time.perf_counter()

        def dec(x):
            with torch.no_grad():
                return vae.decode((1 / sf * x.permute(0, 2, 1, 3, 4)).float()).sample.to(torch.bfloat16)
        rgb = dec(lat[:, :, :16])
        rgb = (rgb[0].permute(1, 0, 2, 3) / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float()
        disp = dec(lat[:, :, 16:32]).mean(dim=1)
        disp = torch.square(disp * 0.5 + 0.5).float()[0]
        res[task] = {"rgb_psnr_db": fc.psnr(rgb[:, ::s, ::s], torch.from_numpy(z["rgb_s8"].astype(np.float32))),
                     "disparity_rel_l2": fc.metrics(disp[:, ::s, ::s], torch.from_numpy(z["disparity_s8"].astype(np.float32)))["rel_l2"],
                     "final_latents_rel_l2": fc.metrics(lat.float(), fc.from_bf16_bits(z["final_latents_bits"]).float())["rel_l2"], "seconds_cpu": time.perf_counter() - t0}
        print(task, json.dumps(res[task]), flush=True)
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
