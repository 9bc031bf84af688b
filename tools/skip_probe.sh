# usage (GPU box): bash tools/skip_probe.sh <workload> [bench args]   -- step time with one kernel family dropped at a time
# (tools/skip_probe_build.sh built graphtrans_amd/libgt_skip.so).  Reads as: how much of the step hangs on that family.
W=$1; shift
export GT_LIB_PATH=$PWD/graphtrans_amd/libgt_skip.so
for k in none k_ln_bwd_d128 k_ln_bwd_d128,k_ln_bwd_finish k_seq_gather,k_seq_scatter k_xent k_heads_dx,k_heads_dx_reduce k_heads_fwd k_small_dw k_bn_stats_partial,k_bn_stats_finish k_bn_apply k_bn_bwd_partial,k_bn_bwd_finish k_bn_bwd_apply k_agg_reduce k_split_reduce k_segsum,k_bcast_add k_small_fwd,k_small_dx,k_bn_small k_lin3_dw k_dw16 k_attn k_aggw_fwd k_aggw_bwd k_adamw k_eseg,k_esort k_lin1 "k_lin3<" k_lin3r_dw "k_lin3r<" k_segsum_chunks,k_segsum_fixup k_count,k_scan,k_fill,k_sort,k_gather,k_w3_image,k_w1_image k_embed_fwd k_small_fwd,k_bn_small_fwd k_small_dx,k_bn_small_bwd none; do
  export GT_SKIP=$k
  python bench.py --workload $W --steps 60 --warmup 10 --no-kernel-timing --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %.4f ms' % ('$k', d['ms_per_step']))"
done
