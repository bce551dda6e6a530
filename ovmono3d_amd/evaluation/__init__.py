from .omni3d_evaluation import Omni3DEvaluator, inference_on_dataset, instances_to_coco_json  # noqa: F401
from .omni3d_eval import (OMNI3D_ALL, OMNI3D_IN, OMNI3D_OUT, Omni3Deval, Omni3DParams, box3d_overlap, collective_summary,  # noqa: F401
                          evaluate_omni3d, omni3d_json_to_gt)
from .omni3d_gt import (CategoryMap, Omni3DGroundTruth, annotation_ignored, eval_filter_settings,  # noqa: F401
                        filter_settings_from_cfg, ground_truth_records)
from .nhd import cuboid_corners, disentangled_nhd, hungarian_distance  # noqa: F401
