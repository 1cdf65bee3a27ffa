"""``models`` registry with the reference's surface (``/root/reference/models/__init__.py``):
``models.__dict__[args.arch](**kwargs)`` as the training/eval scripts do
(training/train_rig.py:83, train_skin.py:83, train_corr_pose.py:152)."""
from .corrnet import *     # noqa: F401,F403
from .rignet import *      # noqa: F401,F403
from .deformnet import *   # noqa: F401,F403
