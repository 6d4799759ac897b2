"""Mirror of the reference's ``rangedet.core`` for the inference path: ``input`` (the transform classes the config imports,
backed by the fused device kernel rd_input_transform) and ``detection_metric`` (the scalar metric the config instantiates)."""
