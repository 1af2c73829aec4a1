"""Drop-in for the reference's `meshnet` package: the modules on the hot path only
(graph_network, meshnet_network, cloth_network, model_utils)."""
