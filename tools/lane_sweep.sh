for mv in "" "--move=up43:2" "--move=cos.3:2" "--move=t22.3:2" "--move=rep0:3" "--move=wf:1" "--move=wf:3" "--move=knn_down:1" "--move=i3_refine:2" "--move=mfa_proj.1:2" "--move=mfa_proj.2:1" "--move=cos.1:3" "--move=i3_01:3" "--move=swap_f.1:3,swap_f.2:3" "--move=self_search:3" "--move=refine_fps:0" "--move=rep0:-" "--move=wf:-" "--move=knn_down:-" "--move=t22.3:-"; do
  echo "== $mv"; timeout -k 5 100 python tools/lane_order.py $mv - 2>&1 | grep "ms/step"
done
