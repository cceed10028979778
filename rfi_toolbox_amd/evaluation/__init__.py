"""Drop-in for ``rfi_toolbox.evaluation`` metrics (reference evaluation/metrics.py:25-172)."""
from .metrics import (compute_dice, compute_f1, compute_iou, compute_precision, compute_recall,
                      confusion_counts, evaluate_segmentation)

__all__ = ["compute_iou", "compute_precision", "compute_recall", "compute_f1", "compute_dice",
           "evaluate_segmentation", "confusion_counts"]
