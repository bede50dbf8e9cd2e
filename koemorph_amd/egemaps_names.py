"""Names of the 88 eGeMAPSv02 functionals in openSMILE's output order (what `smile.feature_names` lists for
FeatureSet.eGeMAPSv02 / FeatureLevel.Functionals); km_egemaps_functionals writes its outputs in this order."""
from typing import List

FEATURE_NAMES: List[str] = (
    [f"F0semitoneFrom27.5Hz_sma3nz_{s}" for s in ("amean", "stddevNorm", "percentile20.0", "percentile50.0", "percentile80.0",
                                                   "pctlrange0-2", "meanRisingSlope", "stddevRisingSlope", "meanFallingSlope",
                                                   "stddevFallingSlope")] +
    [f"loudness_sma3_{s}" for s in ("amean", "stddevNorm", "percentile20.0", "percentile50.0", "percentile80.0", "pctlrange0-2",
                                     "meanRisingSlope", "stddevRisingSlope", "meanFallingSlope", "stddevFallingSlope")] +
    ["spectralFlux_sma3_amean", "spectralFlux_sma3_stddevNorm"] +
    [f"mfcc{i}_sma3_{s}" for i in (1, 2, 3, 4) for s in ("amean", "stddevNorm")] +
    [f"{n}_sma3nz_{s}" for n in ("jitterLocal", "shimmerLocaldB", "HNRdBACF", "logRelF0-H1-H2", "logRelF0-H1-A3",
                                  "F1frequency", "F1bandwidth", "F1amplitudeLogRelF0", "F2frequency", "F2bandwidth",
                                  "F2amplitudeLogRelF0", "F3frequency", "F3bandwidth", "F3amplitudeLogRelF0")
     for s in ("amean", "stddevNorm")] +
    [f"{n}V_sma3nz_{s}" for n in ("alphaRatio", "hammarbergIndex", "slope0-500", "slope500-1500", "spectralFlux",
                                   "mfcc1", "mfcc2", "mfcc3", "mfcc4") for s in ("amean", "stddevNorm")] +
    [f"{n}UV_sma3nz_amean" for n in ("alphaRatio", "hammarbergIndex", "slope0-500", "slope500-1500", "spectralFlux")] +
    ["loudnessPeaksPerSec", "VoicedSegmentsPerSec", "MeanVoicedSegmentLengthSec", "StddevVoicedSegmentLengthSec",
     "MeanUnvoicedSegmentLength", "StddevUnvoicedSegmentLength", "equivalentSoundLevel_dBp"])
assert len(FEATURE_NAMES) == 88
