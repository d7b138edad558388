"""Top-level alias package: the import paths Vidu4D uses (`gs.gaussian_renderer`, `gs.scene.cameras`,
`gs.scene.gaussian_model`, `gs.utils.*`) resolve to the MI355X-native implementations in vidu4d_amd.gs."""
