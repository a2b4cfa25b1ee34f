// mds_platform_rt.h (test simulator): hipemu.h is force-included by the simulator build; nothing to add.
#pragma once
