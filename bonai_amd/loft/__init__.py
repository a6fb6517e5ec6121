"""MI355X-native mirror of the reference's LOFT/FOA model surface (mmdet.models / mmdet.core names)."""
from . import backbone, core, detector, hrnet, losses, roi, rpn  # noqa: F401  (registers every class)
from .builder import (ANCHOR_GENERATORS, BACKBONES, BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, DETECTORS, HEADS,  # noqa
                      IOU_CALCULATORS, LOSSES, NECKS, ROI_EXTRACTORS, SHARED_HEADS, build_anchor_generator,
                      build_assigner, build_backbone, build_bbox_coder, build_detector, build_head, build_loss,
                      build_neck, build_roi_extractor, build_sampler)
from .detector import LOFT  # noqa: F401
