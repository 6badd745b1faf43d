"""Material / scene constants: the numerical contract of fluidlab/configs/macros.py, restated as one table."""
import numpy as np

# ---- material ids (macros.py:1-17) and scene-object ids (macros.py:19-35)
_MATERIAL_IDS = dict(WATER=0, MILK=1, COFFEE=2, ELASTIC=3, ICECREAM=4, RIGID=5, RIGID_HEAVY=6, RIGID_LIGHT=7, MILK_VIS=8,
                     COFFEE_VIS=9, ELASTIC_DEMO=10, PLASTIC_DEMO=11, INVISCID_DEMO=12, VISCOUS_DEMO=13, INVISCID_DEMO2=14,
                     INVISCID_DEMO3=15, ICECREAM1=16)
_OBJECT_IDS = dict(CUP=50, TANK=51, LADDLE=52, POURER=53, DISPENSER=54, CONE=55, ROBOT=56, BOTTLE=57, PILLAR=58, STIRRER=59,
                   PLATE=60, BOWL=61, FRAME=100, TARGET=101, EFFECTOR=102)
globals().update(_MATERIAL_IDS)
globals().update(_OBJECT_IDS)

# ---- material classes (macros.py:37-41)
MAT_LIQUID, MAT_PLASTO_ELASTIC, MAT_ELASTIC, MAT_RIGID, MAT_PLASTO_ELASTIC_DEMO = 200, 201, 202, 203, 204

# name, class, mu, lambda, rho, rgba   (macros.py:45-201)
_WHITE, _RED, _BLUE, _PINK = (1.0, 1.0, 1.0, 1.0), (1.0, 0.2, 0.1, 1.0), (0.3, 0.8, 1.0, 1.0), (1.0, 0.5, 0.5, 1.0)
_TABLE = {
    'WATER':          ('water',          MAT_LIQUID,              0.0,    277.78, 1.0,  _BLUE),
    'INVISCID_DEMO':  ('inviscid-demo',  MAT_LIQUID,              0.0,    277.78, 5.0,  _BLUE),
    'INVISCID_DEMO2': ('inviscid-demo2', MAT_LIQUID,              0.0,    277.78, 1.0,  _RED),
    'INVISCID_DEMO3': ('inviscid-demo3', MAT_LIQUID,              0.0,    277.78, 3.0,  _RED),
    'VISCOUS_DEMO':   ('viscous-demo',   MAT_LIQUID,              800.0,  277.78, 5.0,  _RED),
    'MILK':           ('milk',           MAT_LIQUID,              0.0,    277.78, 0.5,  (0.9, 0.9, 0.9, 1.0)),
    'COFFEE':         ('coffee',         MAT_LIQUID,              0.0,    277.78, 1.0,  (0.58, 0.42, 0.22, 1.0)),
    'MILK_VIS':       ('milk-viscous',   MAT_LIQUID,              200.0,  277.78, 1.0,  (0.9, 0.9, 0.9, 1.0)),
    'COFFEE_VIS':     ('coffee-viscous', MAT_LIQUID,              200.0,  277.78, 1.0,  (0.58, 0.42, 0.22, 1.0)),
    'ELASTIC':        ('elastic',        MAT_ELASTIC,             416.67, 277.78, 1.0,  _WHITE),
    'ELASTIC_DEMO':   ('elastic-demo',   MAT_ELASTIC,             10.0,   100.0,  1.0,  _WHITE),
    'PLASTIC_DEMO':   ('plastic-demo',   MAT_PLASTO_ELASTIC_DEMO, 160.0,  277.78, 1.0,  _WHITE),
    'ICECREAM':       ('ice-cream',      MAT_PLASTO_ELASTIC,      416.67, 277.78, 0.5,  _WHITE),
    'ICECREAM1':      ('ice-cream1',     MAT_PLASTO_ELASTIC,      216.67, 277.78, 0.5,  _WHITE),
    'RIGID':          ('rigid',          MAT_RIGID,               416.67, 277.78, 1.0,  _PINK),
    'RIGID_HEAVY':    ('rigid-heavy',    MAT_RIGID,               416.67, 277.78, 10.0, _PINK),
    'RIGID_LIGHT':    ('rigid-light',    MAT_RIGID,               416.67, 277.78, 0.5,  _PINK),
}
MAT_NAME = {_MATERIAL_IDS[k]: v[0] for k, v in _TABLE.items()}
MAT_CLASS = {_MATERIAL_IDS[k]: v[1] for k, v in _TABLE.items()}
MU = {_MATERIAL_IDS[k]: v[2] for k, v in _TABLE.items()}
LAMDA = {_MATERIAL_IDS[k]: v[3] for k, v in _TABLE.items()}
RHO = {_MATERIAL_IDS[k]: v[4] for k, v in _TABLE.items()}
COLOR = {_MATERIAL_IDS[k]: v[5] for k, v in _TABLE.items()}
COLOR.update({CUP: (0.9, 0.9, 0.9, 1.0), TANK: (0.70, 0.95, 0.96, 0.6), BOWL: (0.78, 0.56, 0.12, 1.0), LADDLE: _WHITE,
              POURER: _WHITE, DISPENSER: _WHITE, CONE: (0.645, 0.474, 0.303, 1.0), ROBOT: _WHITE,
              BOTTLE: (0.70, 0.95, 0.96, 0.5), PILLAR: _WHITE, STIRRER: _WHITE, PLATE: _WHITE,
              FRAME: (1.0, 0.2, 0.2, 1.0), TARGET: (0.2, 0.9, 0.2, 0.4), EFFECTOR: (1.0, 0.0, 0.0, 1.0)})
FRICTION = {CUP: 0.5, TANK: 0.5, BOWL: 0.0, LADDLE: 0.1, CONE: 8.0, BOTTLE: 0.1, PILLAR: 0.0, STIRRER: 8.0, PLATE: 0.1}

# ---- precision (macros.py:207-213): the engine is fp32; DTYPE_NP is what crosses the C ABI
DTYPE_NP = np.float32
EPS = 1e-12
NOWHERE = [-100.0, -100.0, -100.0]          # macros.py:216
