"""model/Parameters.py of the reference, constant for constant (model/Parameters.py:6-49), without the Keras/TF imports."""
import os

import numpy as np
import torch

backbone_name = "resnet50"

# Paths for testing
root_dir = os.getcwd()
trained_model_path = os.path.join(root_dir, 'models', 'training-U7U2ycFZg_resnet50_48.h5')

# Pre-processing
image_min_side = 800
image_max_size = 1333
image_scaling_factor = 127.5
image_subtraction_factor = 1.

# Training parameters
steps_per_epoch = 2000
num_epochs = 100
learning_rate = 1e-4
batch_size = 1
multi_gpu = torch.cuda.device_count()          # reference: len(tf.config.list_physical_devices('GPU'))
multiprocessing = False
num_workers = 1
max_queue_size = 10

# Anchor parameters
sizes = [32, 64, 128, 256, 512]
strides = [8, 16, 32, 64, 128]
ratios = np.array([0.5, 1, 2], 'float32')
scales = np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)], 'float32')

# Miscellaneous
negative_overlap = 0.4
positive_overlap = 0.5
std = [0.2, 0.2, 0.2, 0.2]
class_mapping = {"table": 0}
