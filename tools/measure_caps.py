import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, warnings
warnings.filterwarnings('ignore')
from oracle import refshim
refshim.install()
import metaworld
from metaworld.policies import ENV_POLICY_MAP
from metaworld.env_dict import ALL_V3_ENVIRONMENTS
rng=np.random.default_rng(0)
mc=me=0
for name in ALL_V3_ENVIRONMENTS:
    mt1 = metaworld.MT1(name, seed=42)
    env = mt1.train_classes[name]()
    policy = ENV_POLICY_MAP[name]()
    for k,task in enumerate(mt1.train_tasks[:3]):
        env.set_task(task); obs,_=env.reset()
        for step in range(300):
            a = policy.get_action(obs) if k<2 else rng.uniform(-1,1,4)
            a = np.clip(a + (rng.normal(0,0.3,4) if k==1 else 0), -1,1)
            obs, r, te, tr, info = env.step(a)
    i=env.data._od.info()
    print(f"{name:30s} max_ncon {i['max_ncon']:3d} max_nefc {i['max_nefc']:3d} ngeom {env.model.ngeom} npair {len(env.model._src.arrays['pair_geom'])}", flush=True)
    mc=max(mc,i['max_ncon']); me=max(me,i['max_nefc'])
print(mc,me)
