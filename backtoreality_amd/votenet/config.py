"""Dataset configs for the harness.  The reference's configs
(scannet/model_util_scannet.py:71-87 ScannetDatasetConfig_md40: 22 classes / 1 heading bin /
22 size clusters; matterport/model_util_matterport.py:16-30: 13 / 12 / 13) load their mean box
sizes from dataset meta-data that is not available offline, so the harness draws a
deterministic stand-in table of plausible furniture sizes instead."""
import numpy as np


class DatasetConfig(object):
    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr=None,
                 max_num_obj=64, seed=7):
        self.num_class = int(num_class)
        self.num_heading_bin = int(num_heading_bin)
        self.num_size_cluster = int(num_size_cluster)
        self.max_num_obj = int(max_num_obj)
        if mean_size_arr is None:
            rng = np.random.default_rng(seed)
            mean_size_arr = rng.uniform(0.3, 1.8, size=(self.num_size_cluster, 3))
        self.mean_size_arr = np.asarray(mean_size_arr, dtype=np.float64)
        assert self.mean_size_arr.shape == (self.num_size_cluster, 3)


def scannet_md40():
    """Shape of ScannetDatasetConfig_md40 (MAX_NUM_OBJ 64, scannet_detection_dataset.py:26)."""
    return DatasetConfig(22, 1, 22, max_num_obj=64)


def matterport_md40():
    """Shape of MatterportDatasetConfig_md40 (MAX_NUM_OBJ 256)."""
    return DatasetConfig(13, 12, 13, max_num_obj=256)
