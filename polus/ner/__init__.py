"""polus.ner -> polus_amd.ner (re-export)."""
