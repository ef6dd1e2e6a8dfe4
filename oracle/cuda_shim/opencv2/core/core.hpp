// stand-in for <opencv2/core/core.hpp>: kfusion/src/internal.hpp includes it but the device-side declarations use none of it.
// TEST INFRASTRUCTURE ONLY (oracle/_ref build).
#pragma once
