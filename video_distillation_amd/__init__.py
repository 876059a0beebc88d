"""MI355X-native hot path of yuz1wan/video_distillation (see README.md / DESIGN.md)."""
import os as _os

# The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialise.
# The gradient-matching trainer runs its class terms on 8 lane streams (distill.GMTrainer): with 4 queues they
# run 3.0 steps/s, with 16 queues 3.5.  Read once when the runtime initialises, so it has to be in the environment before
# the first HIP call of the process -- importing this package is normally early enough; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
