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
        # ScanNet boxes are axis aligned: class2angle is constant 0 there
        # (scannet/model_util_scannet.py:102-106); with heading bins it is the bin centre plus
        # the residual (matterport/model_util_matterport.py:51-62).
        self.axis_aligned = self.num_heading_bin == 1

    def class2angle(self, pred_cls, residual, to_label_format=True):
        """Inverse of angle2class for one box (scalars), float64."""
        if self.axis_aligned:
            return 0
        angle = pred_cls * (2 * np.pi / float(self.num_heading_bin)) + residual
        if to_label_format and angle > np.pi:
            angle = angle - 2 * np.pi
        return angle

    def class2size(self, pred_cls, residual, ratio=1.0):
        """Inverse of size2class: mean size of the cluster + residual (l, w, h), float64."""
        return (self.mean_size_arr[pred_cls, :] + residual) * ratio

    def class2angle_batch(self, pred_cls, residual):
        """class2angle over tensors: (…) int64, (…) float -> (…) float64."""
        import torch
        if self.axis_aligned:
            return torch.zeros(pred_cls.shape, dtype=torch.float64, device=pred_cls.device)
        angle = pred_cls.double() * (2 * np.pi / float(self.num_heading_bin)) + residual.double()
        return torch.where(angle > np.pi, angle - 2 * np.pi, angle)

    def class2size_batch(self, pred_cls, residual):
        """class2size over tensors: (…) int64, (…, 3) float -> (…, 3) float64."""
        import torch
        # (the table goes to the device once, not per call: a pageable host-to-device copy is
        # a 0.6 ms synchronous round trip, two per evaluation batch)
        held = self.__dict__.setdefault('_mean_size_on', {}).get(pred_cls.device)
        if held is None or not np.array_equal(held[0], self.mean_size_arr):
            host = np.array(self.mean_size_arr, copy=True)
            held = (host, torch.from_numpy(host).to(pred_cls.device))
            self._mean_size_on[pred_cls.device] = held
        return held[1][pred_cls] + residual.double()


def scannet_md40():
    """Shape of ScannetDatasetConfig_md40 (MAX_NUM_OBJ 64, scannet_detection_dataset.py:26)."""
    return DatasetConfig(22, 1, 22, max_num_obj=64)


def matterport_md40():
    """Shape of MatterportDatasetConfig_md40 (MAX_NUM_OBJ 256)."""
    return DatasetConfig(13, 12, 13, max_num_obj=256)
