for v in 1 0 1 0; do
KP2D_TR=$v python3 tools/layer_profile.py --reps 5 2>/dev/null > gpurun_out/lp_tr$v.txt
done
paste <(awk '{print $1, $2, $3}' gpurun_out/lp_tr1.txt) <(awk '{print $3}' gpurun_out/lp_tr0.txt)
