"""MI355X-native deformable-voxel physics step (drop-in for Impact hot path)."""
