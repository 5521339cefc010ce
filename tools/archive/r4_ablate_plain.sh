cd $GRAFT_REPO_ROOT
for L in "encoder.model.15.conv 2000" "encoder.model.12.conv 10000" "decoder.model.3.convtr 250" "decoder.model.6.convtr 2000"; do
for a in 0 4 1 2 16 5; do
FC_ABLATE=$a timeout 120 python tools/ablate_layer.py $L 2>&1 | grep "^ablate"
done; done
