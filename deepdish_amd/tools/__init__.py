"""MI355X-native counterparts of the reference's tools package (detector / encoder plugins,
count-line geometry)."""
