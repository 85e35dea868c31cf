#!/usr/bin/env python
"""Run the REFERENCE scripted policies (metaworld/policies) with the REFERENCE env classes on the oracle engine
(oracle/refshim.py) and report success per task: the reference own behavioural gate (tests/.../test_scripted_policies.py:35).
Needs /root/reference.  usage: tools/policy_gate.py [task ...]"""
import sys, time, traceback
sys.path.insert(0,'/root/repo')
import numpy as np, warnings
warnings.filterwarnings('ignore')
np.set_printoptions(precision=4, suppress=True, linewidth=200)
from oracle import refshim
refshim.install()
import metaworld
from metaworld.policies import ENV_POLICY_MAP
from metaworld.env_dict import ALL_V3_ENVIRONMENTS
names = list(ALL_V3_ENVIRONMENTS.keys())
import os
sel = sys.argv[1:] or names
NE = int(os.environ.get('MW_GATE_GOALS', '5'))        # the reference's own test walks all 50 (MW_GATE_GOALS=50)
res = {}
for name in sel:
    t=time.time()
    try:
        mt1 = metaworld.MT1(name, seed=42)
        env = mt1.train_classes[name]()
        policy = ENV_POLICY_MAP[name]()
        succ=0; ov=0; fails=[]
        for k,task in enumerate(mt1.train_tasks[:NE]):
            env.set_task(task)
            obs,info = env.reset()
            for step in range(500):
                a = policy.get_action(obs)
                obs, r, te, tr, info = env.step(a)
                if int(info['success'])==1:
                    succ+=1; break
            else:
                fails.append(k)
            ov = max(ov, env.data._od.info()['overflow'])
        res[name]=succ
        print(f"{name:32s} succ {succ}/{NE}  nv={env.model.nv} overflow={ov} t={time.time()-t:.1f}s failed goals {fails}", flush=True)
    except Exception as e:
        print(f"{name:32s} ERROR {type(e).__name__}: {str(e)[:150]}", flush=True)
        res[name]=-1
print(f'tasks passing the 80 % gate ({NE} goals):', sum(1 for v in res.values() if v >= 0.8 * NE), 'of', len(res))
