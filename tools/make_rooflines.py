"""Per-kernel roofline table of the benchmarked step (BASELINE configs[1]) from a round's committed profile passes:

    python tools/make_rooflines.py <kernel_stats.csv> <steps in the trace> [pmc_traffic.json] [pmc_mfma.json] > profiles/rNN_rooflines.json

For every kernel that takes >= 0.5 % of the summed kernel time: launches per step, average duration, share, the roofline that
bounds it (hbm | mfma | latency), its ALGORITHMIC work per launch (bytes or FLOPs from the shapes of configs[1]: B = 32,
T' = 376, N = B T' = 12 032 frames, cells = B T' (U+1) = 1 275 392, d = 256, H = 640, V = 257 -> 264 columns, 17.3 M trainable
parameters), the achieved rate and its fraction of the datasheet peak (8 TB/s HBM3E, 2.5 PFLOP/s dense bf16/f16 MFMA:
/opt/skills/guides/MI355X_MICROARCH.md).  PMC columns (HBM bytes per launch, matrix-pipe busy share) are attached when the
counter passes are given.  bench.py embeds the table as `rooflines`.
"""
import csv
import json
import sys

HBM, MFMA = 8000.0, 2500.0          # GB/s, TFLOP/s
B, T, U1, d, H, LD = 32, 376, 106, 256, 640, 264
N, CELLS, NPARAM = B * T, B * T * U1, 17.3e6   # trainable parameters at freeze_encoder_till = 12 (FlatParams.numel)
LIVE = 0.827                         # share of lattice cells in front of frame T_b + 8 (bench lengths, bench.py touched_fraction)

# substring of the kernel name -> (label, bound, algorithmic work per launch, unit, note)
WORK = [
    ("joint_dh_fused_kernel", ("joint hidden gradient dH = G W + mask + both reductions", "mfma", 2.0 * CELLS * H * LD, "flop", "")),
    ("joint_fwd_kernel", ("joint forward relu(f+g) W^T + loss front end", "mfma", 2.0 * CELLS * H * 257, "flop", "")),
    ("joint_dw_fused_kernel", ("joint weight gradient dW = G^T hidden", "mfma", 2.0 * CELLS * LD * H, "flop", "")),
    ("joint_grad_h_db_kernel", ("transducer gradient, in place over the f16 lattice", "hbm", 2 * 2.0 * CELLS * 257 * LIVE, "byte",
                                "touched cells only (bench.py roofline)")),
    ("ffn_fused_kernel<256, true, false, true>", ("feed-forward module LN -> W1 -> SiLU -> W2 -> residual -> LN + the q|k|v projection as a tail phase",
                                                  "mfma", 2.0 * 2 * N * d * 4 * d + 2.0 * N * d * 3 * d, "flop",
                                                  "paced by 1.4 MB of weights per workgroup through LDS")),
    ("ffn_fused_kernel", ("feed-forward module LN -> W1 -> SiLU -> W2 -> residual", "mfma", 2.0 * 2 * N * d * 4 * d, "flop",
                          "paced by 1 MB of weights per workgroup through LDS")),
    ("gemm_bf16_nt_kernel<128, 128, true>", ("subsampling conv2 as implicit GEMM", "mfma", 2.0 * (B * 376 * 20) * 256 * 2304, "flop", "")),
    ("gemm_bf16_nt_kernel<96, 128", ("subsampling Linear [N x 5120] x [256 x 5120]", "mfma", 2.0 * N * 256 * 5120, "flop", "")),
    ("gemm_bf16_nt_dma_kernel", ("long-K projections (K >= 512: feed-forward W2, data gradients), K-pipelined by LDS-DMA", "mfma", 2.0 * N * 256 * 1024, "flop",
                                 "[12032 x 1024] x [1024 x 256] is the most frequent shape (12 of ~18 launches)")),
    ("gemm_bf16_nt_kernel<64, 256", ("attention out-projection + residual + LayerNorm of the convolution module in one launch", "hbm",
                                     2.0 * N * d + 4.0 * N * d + 4.0 * N * d + 2.0 * N * d, "byte", "1.6 GFLOP; latency-bound at this size")),
    ("gemm_bf16_nt_kernel<64, 128", ("projections of the blocks / heads (K = 256 .. 1024, mixed shapes)", "mfma", 361e9 / 95.0, "flop",
                                     "average over the step's ~95 launches (361 GFLOP per step)")),
    ("relpos_flash_fwd_kernel", ("rel-pos attention forward, key-tiled", "mfma", 7.0e9, "flop", "latency / VALU (exp, band strip) at T' = 376")),
    ("relpos_flash_bwd_q_kernel", ("attention backward, query owner", "mfma", 3 * 7.0e9, "flop", "")),
    ("relpos_flash_bwd_kv_kernel", ("attention backward, key owner", "mfma", 2 * 7.0e9, "flop", "")),
    ("gemm_tn_grouped_kernel", ("weight gradients of a trainable block (grouped TN GEMM)", "mfma", 3 * 37.8e9 / 9.0, "flop",
                                "average over the block's three grouped launches")),
    ("gemm_tn_kernel", ("weight gradients of the heads / LSTM (TN GEMM)", "mfma", 2.0 * N * 640 * 256, "flop", "largest of the mixed shapes")),
    ("gemm_bnsilu_kernel", ("BatchNorm + SiLU + pointwise_conv2", "mfma", 2.0 * N * d * d, "flop", "latency-bound at this size")),
    ("conv1_relu_cl_kernel", ("subsampling conv1 (1 -> 256 channels)", "hbm", 4.0 * B * 80 * 1501 + 2.0 * B * 751 * 40 * 256, "byte", "")),
    ("dwconv_fwd_kernel<31, 2>", ("depthwise conv + BatchNorm sums (frozen blocks)", "hbm", 2.0 * N * d + 4.0 * N * d, "byte", "")),
    ("layernorm_kernel", ("LayerNorm", "hbm", 4.0 * N * d + 2.0 * N * d, "byte", "")),
    ("layernorm_bwd_kernel", ("LayerNorm backward", "hbm", 3 * 4.0 * N * d, "byte", "")),
    ("adamw_seg_kernel", ("AdamW over the flat buffers", "hbm", 28.0 * NPARAM, "byte", "")),
    ("cl_penalty_kernel", ("EWC penalty into .grad", "hbm", 16.0 * NPARAM, "byte", "partly served by the Infinity Cache")),
    ("lstm_fwd_kernel", ("persistent LSTM forward (106 hand-offs)", "latency", 0.0, "", "40 workgroups, side stream")),
    ("lstm_bwd_kernel", ("persistent LSTM backward (106 hand-offs)", "latency", 0.0, "", "40 workgroups, side stream")),
    ("ctc_alpha_beta", ("CTC alpha / beta on raw logits", "latency", 0.0, "", "one wave per (utterance, direction), side stream")),
    ("ctc_grad_logits_kernel", ("CTC gradient -> bf16 GEMM operand", "hbm", 4.0 * N * 264 + 2.0 * N * 264, "byte", "side stream")),
    ("rnnt_alpha_beta", ("transducer alpha / beta wavefront", "latency", 0.0, "", "481 dependent diagonal steps")),
    ("greedy_decode_mfma_kernel", ("greedy transducer decode of the in-step WER (one launch per step)", "latency", 0.0, "",
                                   "side stream, 4 workgroups per utterance; a dependent chain of head evaluations and LSTM steps out of "
                                   "L2 (36 us per emitted symbol, 10 us per 16 frames): the average is carried by the first ~5 steps of the "
                                   "run, where the freshly initialised model emits max_symbols labels at every frame (100+ ms per launch); "
                                   "~2 ms per launch afterwards, under the joint / backward")),
]


def main():
    stats, steps = sys.argv[1], float(sys.argv[2])
    traffic = json.load(open(sys.argv[3]))["kernels"] if len(sys.argv) > 3 else {}
    busy = json.load(open(sys.argv[4]))["kernels"] if len(sys.argv) > 4 else {}
    rows = list(csv.DictReader(open(stats)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    out = []
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        share = float(r["TotalDurationNs"]) / total
        if share < 0.005:
            continue
        name = r["Name"]
        avg_us = float(r["AverageNs"]) / 1e3
        ent = {"kernel": name.replace("(anonymous namespace)::", "").split("(")[0][:70], "launches_per_step": round(int(r["Calls"]) / steps, 1),
               "avg_us": round(avg_us, 1), "share_of_kernel_time": round(share, 4)}
        for key, (label, bound, work, unit, note) in WORK:
            if key in name:
                ent.update(what=label, bound=bound)
                if bound == "mfma" and work:
                    ach = work / (avg_us * 1e-6) / 1e12
                    ent.update(algorithmic_flops_per_launch=int(work), achieved=round(ach, 1), unit="TFLOP/s", peak=MFMA, frac=round(ach / MFMA, 4))
                elif bound == "hbm" and work:
                    ach = work / (avg_us * 1e-6) / 1e9
                    ent.update(algorithmic_bytes_per_launch=int(work), achieved=round(ach, 1), unit="GB/s", peak=HBM, frac=round(ach / HBM, 4))
                if note:
                    ent["note"] = note
                break
        else:
            ent.update(what="(other)", bound=None)
        short = ent["kernel"].split("<")[0].strip().split(" ")[-1]
        for k, v in traffic.items():
            if k in name and isinstance(v, dict) and "hbm_bytes_per_launch" in v:
                ent["pmc_hbm_bytes_per_launch"] = v["hbm_bytes_per_launch"]
                break
        for k, v in busy.items():
            if k in name and isinstance(v, dict):
                for kk in ("mfma_busy_share_of_1024_simds", "mfma_busy_share", "mfma_busy"):
                    if kk in v:
                        ent["pmc_mfma_busy_share"] = v[kk]
                break
        out.append(ent)
    json.dump({"source": stats, "steps_in_trace": steps, "sum_kernel_ms_per_step": round(total / steps / 1e6, 3), "kernels": out},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
