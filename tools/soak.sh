#!/bin/bash
# soak: 3000 iterations of `caffe train` on the cfg-2 example; prints the loss / violations trajectory
cd $GRAFT_REPO_ROOT
sed 's/max_iter: 200/max_iter: 3000/; s/display: 20/display: 300/; s/base_lr: 0.001/base_lr: 0.01/' examples/videovec_cfg2_solver.prototxt > /tmp/soak.prototxt
caffe_facade/build/caffe train --solver=/tmp/soak.prototxt --log_file=/tmp/soak.log > /dev/null 2>&1
grep -E "Iteration [0-9]+, loss|train_violations" /tmp/soak.log | sed 's/^.*\] //' | paste - - | head -14
