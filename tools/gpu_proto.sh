#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/p_base.json
cp curvis_amd/lib/libcurvis_hip.so /tmp/orig.so; cp build/libcurvis_proto.so curvis_amd/lib/libcurvis_hip.so
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/p_proto.json
python - > gpurun_out/p_proto_check.log 2>&1 <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, common, oracle_lib as O, curvis_amd
sp, sn = common.make_skies(512,256,"smooth")
om, oc, pm, pc = common.scene("ellis", res=(256,144))
ctx = curvis_amd.Context(0)
s = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc, context=ctx)
got = s.render_image(4096, 100.0, 0.05)
want, _, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05)
d = np.abs(got.astype(int)-want.astype(int)).max(axis=2)
print("proto vs oracle: exact %.4f  <=1 %.4f  steps %d vs %d" % ((d==0).mean(), (d<=1).mean(), s.last_stats.steps, st.steps))
PY
cp /tmp/orig.so curvis_amd/lib/libcurvis_hip.so
