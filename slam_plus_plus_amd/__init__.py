"""MI355X-native drop-in for the Lambda-solve hot path of SLAM++."""
