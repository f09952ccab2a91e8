#!/bin/bash
# The build's counterpart of the reference's tools/test_vo_scene.sh (lines 1-11): for every VKITTI2 scene, first the VO run that
# writes the trajectory and prints the ATE-RMSE (test_vo.py, panoptic filter on), then the per-frame flow / depth pass for the
# panoptic half (test_vo2.py).  Same two commands, same arguments, same directory layout; the drivers are tools/test_vo.py and
# tools/test_vo2.py of this repository (pvo_amd on libpvo_hip.so instead of droid_slam on the CUDA extensions).
#   bash tools/test_vo_scene.sh [--weights checkpoints/vkitti2_dy_train_semiv4_080000.pth]   (extra arguments go to test_vo.py)
# Needs the dataset under datasets/Virtual_KITTI2/<scene> (not in this image: the synthetic drivers are tools/test_vo.py --help).
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
for scene in 'Scene01' 'Scene02' 'Scene06' 'Scene18' 'Scene20'
do
    # poses + rmse
    python "$HERE/test_vo.py" \
    --datapath=datasets/Virtual_KITTI2/$scene \
    --disable_vis --segm_filter True "$@"

    # flow and depth, frame by frame
    python "$HERE/test_vo2.py" --scene $scene
done
